// Fused "gradient through image_text_input into the adapter" kernel of the ARPDT train step (16-bit modes).
//
//   dY[R, Kin]    = dz[R, E] . Wi[E, Kin]                      (gradient w.r.t. the adapter's mixed output y, arp_dt/ARPDT.py:476-484)
//   dApre[R, Kin] = res * dY * (A > 0)                          (y = res * A + (1 - res) * x with A = relu(.), ARPDT.py:462-472)
//   colpart       = column sums of the rounded dApre per (row block, token)  -> the Dense_1 bias gradient after one colsum_kernel
//   dres_part     = sum dY * (A - x) per workgroup                           -> d loss / d res after one reduce_sum_kernel
//
// It replaces three launches of round 1 / early round 2 -- the transposed operand shadow of Wi (refresh_shadows: 50 MB read + 50 MB
// written per step), the skinny NT GEMM that wrote dY (50 MB) and the masked copy that read it back with A and x -- by one pass that
// reads Wi as it lies in memory ([E, Kin] rows, fetched k-major with the CDNA4 transposing LDS read as gemm_tn.h does), A and x once,
// and writes dApre once: 251 MB instead of 501 MB of HBM traffic and one launch instead of three (0.125 ms -> see DESIGN.md section 6).
//
// Requirements: E a multiple of 32 and <= 128; D (= enc_dim, the adapter width) and Kin multiples of 128; Kin a multiple of D.
#pragma once
#include "common.h"

namespace arp {

struct AdapterDyArgs {
    const void* dz;     // [>= R, E] T, already scaled by the caller's power-of-two activation scale
    const void* Wi;     // [E, Kin] T: image_text_input/kernel in device layout [out, in], operand-type copy
    const void* A;      // [R, Kin] T: the adapter MLP's output (post-ReLU), = [R * tokens, D]
    const float* x32;   // [R, Kin] f32: the stop-gradient encodings
    const void* x16 = nullptr;  // ... or their operand-type copy (read instead of x32 when set: half the bytes; arp_dt.hip ARP_DT_DY_X16)
    const float* rw;    // device scalar residual_weight (res = sigmoid(rw) is formed in the kernel)
    void* dApre;        // [R, Kin] T out
    float* colpart;     // [row_blocks * tokens, D] f32 out
    float* dres_part;   // [row_blocks * Kin / 128] f32 out
    int R, E, Kin, D;
};

bool adapter_dy_supported(int E, int D, long long Kin);
inline int adapter_dy_row_blocks(int R) { return (R + 127) / 128; }
// tcode: 1 = bf16, 2 = f16
int launch_adapter_dy(int tcode, const AdapterDyArgs& g, hipStream_t stream);

}  // namespace arp
