// Internal (C++) hook that lets the policy step run the frozen encoder on its own stream.
#pragma once
#include <hip/hip_runtime.h>

struct arp_enc;
namespace arp {
// images_dev: f32 NHWC [n, res, res, 3] in HBM; out_dev: f32 [n * tokens, width].  Enqueued on `stream`.
int enc_forward_on(arp_enc* e, hipStream_t stream, const float* images_dev, int n, float* out_dev);
int enc_geometry(arp_enc* e, int* tokens, int* width, int* img_res, int* device);
}  // namespace arp
