// Coset-merged DIT family of the register-radix NTT digit pass (ntt_fast_kernel.cuh): all 2^rate_bits coset transforms of an LDE in one launch per pass
#include "ntt_fast_kernel.cuh"

bool zk_ntt_fast_launch_dit_cs(zk_ctx *ctx, const PassParams &p, bool row, const char *name) { return ntt_fast_launch_family<true, true>(ctx, p, row, name); }
