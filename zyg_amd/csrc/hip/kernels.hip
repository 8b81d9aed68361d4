// One translation unit of the wavefront kernels: compiled once per ZYG_TU (0 .. ZYG_NUM_UNITS-1) and arithmetic mode
// (zyg_amd/build.py), it instantiates the entries of kernel_table.hpp assigned to that unit and exports two C entries:
//   zyg_unit<k>[f]_launch(id, blocks, stream, params, arg)   launch kernel `id` (must belong to this unit) on `stream`
//   zyg_unit<k>[f]_set(which, src, bytes, stream)            set this unit's copy of a __constant__ block
// ZYG_FAST builds (suffix f) are the relaxed-arithmetic instantiations (dmath.hpp): the build renames the device namespace
// (-Dzd=zdf) so that both modes' kernels live side by side in one library.
#include <hip/hip_runtime.h>

#ifndef ZYG_TU
#error "compile with -DZYG_TU=<unit>"
#endif
#if 4 == ZYG_TU && !defined(ZYG_LDS_STACK)
#define ZYG_LDS_STACK 0  // the persistent kernels' unit: their LDS holds the path states, the wide walks' stack stays in scratch
#endif

#ifndef ZYG_TS_BLOCK_ALL
#define ZYG_TS_BLOCK_ALL 0  // 1: every wf_trace_shade instantiation without a wide BLAS runs ZYG_TS_BLOCK threads (wavefront.hpp)
#endif

#include "kernel_table.hpp"
#include "wavefront.hpp"

#define ZYG_CAT3_(a, b, c) a##b##c
#define ZYG_CAT3(a, b, c) ZYG_CAT3_(a, b, c)
#if ZYG_FAST
#define ZYG_UNIT_SYM(name) ZYG_CAT3(zyg_unit, ZYG_TU, f_##name)
#else
#define ZYG_UNIT_SYM(name) ZYG_CAT3(zyg_unit, ZYG_TU, _##name)
#endif

namespace zd {

template <int UNIT>
static int launchIn(unsigned id, unsigned blocks, hipStream_t stream, const Params& P, uint32_t arg) {
    (void)arg;
    const dim3 grid(blocks), block(256);
    const dim3 persistent_block(kPersistentThreads);
    // wf_trace_shade's block size follows its instantiation (wavefront.hpp traceShadeBlock); `blocks` counts 256 entries each
    constexpr uint32_t kTsU = traceShadeBlock<true, false>(), kTsUW = traceShadeBlock<true, ZYG_WIDE(true, false)>();
    const dim3 ts_u_grid((blocks * 256u + kTsU - 1u) / kTsU), ts_u_block(kTsU), ts_uw_grid((blocks * 256u + kTsUW - 1u) / kTsUW), ts_uw_block(kTsUW);
    switch (id) {
#define ZYG_X(unit, kid, kernel, args)                                                       \
    case kid:                                                                                \
        if constexpr (unit == UNIT) {                                                        \
            constexpr bool ts_u = K_TRACE_SHADE_PLAIN_FIRST_U == kid || K_TRACE_SHADE_PLAIN_U == kid ||              \
                                  (ZYG_TS_BLOCK_ALL && kid >= K_TRACE_SHADE_STATS_FIRST && kid <= K_TRACE_SHADE_PLAIN);      \
            constexpr bool ts_uw = K_TRACE_SHADE_PLAIN_FIRST_UW == kid || K_TRACE_SHADE_PLAIN_UW == kid;             \
            constexpr bool persistent = kid >= K_PERSISTENT_LAMBERT && kid <= K_PERSISTENT_SUBST_W;                  \
            hipLaunchKernelGGL(kernel, ts_u ? ts_u_grid : ts_uw ? ts_uw_grid : grid,                                 \
                               ts_u ? ts_u_block : ts_uw ? ts_uw_block : persistent ? persistent_block : block, 0, stream, ZYG_UNPAREN args); \
            return 0;                                                                        \
        } else {                                                                             \
            return -1;                                                                       \
        }
#define ZYG_UNPAREN(...) __VA_ARGS__
        ZYG_KERNEL_TABLE(ZYG_X)
#undef ZYG_X
#undef ZYG_UNPAREN
        default: return -1;
    }
}

}  // namespace zd

extern "C" {

int ZYG_UNIT_SYM(launch)(unsigned id, unsigned blocks, void* stream, const void* params, unsigned arg) {
    return zd::launchIn<ZYG_TU>(id, blocks, static_cast<hipStream_t>(stream), *static_cast<const zd::Params*>(params), arg);
}

int ZYG_UNIT_SYM(set)(int which, const void* src, size_t bytes, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t  e = hipErrorInvalidValue;
    if (ZYG_CONST_VIEW == which && sizeof(zd::View) == bytes)
        e = hipMemcpyToSymbolAsync(HIP_SYMBOL(zd::c_view), src, bytes, 0, hipMemcpyHostToDevice, s);
    else if (ZYG_CONST_SCENE == which && sizeof(zd::DScene) == bytes)
        e = hipMemcpyToSymbolAsync(HIP_SYMBOL(zd::c_scene), src, bytes, 0, hipMemcpyHostToDevice, s);
    else if (ZYG_CONST_SOBOL_LUT == which && sizeof(void*) == bytes)
        e = hipMemcpyToSymbolAsync(HIP_SYMBOL(zd::sobol::c_sobol_lut), src, bytes, 0, hipMemcpyHostToDevice, s);
    return hipSuccess == e ? 0 : -int(e);
}

}  // extern "C"
