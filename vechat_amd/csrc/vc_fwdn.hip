// k_fwdn (several sequences of a window per forward wave, re-alignment rounds: vc_kernels.h) in a translation unit of its own.
//
// An experiment (VC_MULTI=2 / 4; vc_ctx::multi): bit-identical to k_fwd, slower on the job.  Its own translation unit keeps the build of
// vc_api.hip where it was; the options are the same (__graft_entry__.build()).  What looked like a miscompilation of this kernel under
// -structurizecfg-skip-uniform-regions (wrong matrices for a different set of windows from run to run) was a hardware hazard in the
// band store's asm block -- a store of more than 64 bits followed within two wait states by a VALU write of its data registers, which
// the compiler cannot see inside an asm block (vc_kernels.h, "the s_nop") -- that only this body's back-to-back stores ran into.
#define VC_KL static
#include "vc_kernels.h"

// -> 0 launched, 1 no instantiation for this width class.  (Internal to the library: not part of include/vechat_hip.h.)
extern "C" __attribute__((visibility("hidden"))) int vc_launch_fwdn(uint32_t cpl, uint32_t ns_w, uint32_t grid, void* stream, const VcFwdArgs* a) {
    hipStream_t st = static_cast<hipStream_t>(stream);
#define VC_FN(C) if (cpl == C) { if (ns_w == 32) hipLaunchKernelGGL((k_fwdh<C, VC_RING_PRUNED_N>), dim3(grid), dim3(64), 0, st, *a);  /* two sequences on 32 lanes each */ \
                                 else if (ns_w == 4) hipLaunchKernelGGL((k_fwdn<C, VC_RING_PRUNED_N, 4>), dim3(grid), dim3(64), 0, st, *a); \
                                 else hipLaunchKernelGGL((k_fwdn<C, VC_RING_PRUNED_N, 2>), dim3(grid), dim3(64), 0, st, *a); return 0; }
    VC_FN(8) VC_FN(10)
#ifndef VC_FAST_BUILD
    VC_FN(6) VC_FN(12)
#endif
#undef VC_FN
    return 1;
}
