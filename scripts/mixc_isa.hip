// Two MIXC instances of gemm256_nt_kernel alone in a translation unit (8 s to compile): input of scripts/isa_loops.py.
#include "gemm256.h"
namespace arp {
static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }
void set_error(const std::string& m) { g_err = m; }
int launch_gemm2w_dyn(int, int, int, int, const GemmArgs&, hipStream_t) { return -1; }
bool gemm2w_has(int, int, int, int) { return false; }
}
using namespace arp;
int go(const GemmArgs& g) {
    return launch_gemm256_nt<f16_t, f16_t, ACT_NONE, false, 6, false, 1, true>(g, nullptr) + launch_gemm256_nt<f16_t, float, ACT_NONE, true, 6, false, 1, true>(g, nullptr);
}
