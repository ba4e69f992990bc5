// DIT family of the register-radix NTT digit pass (ntt_fast_kernel.cuh)
#include "ntt_fast_kernel.cuh"

bool zk_ntt_fast_launch_dit(zk_ctx *ctx, const PassParams &p, bool row, const char *name) { return ntt_fast_launch_family<true, false>(ctx, p, row, name); }
