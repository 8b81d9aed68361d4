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
#if (4 == ZYG_TU || 5 == ZYG_TU) && !defined(ZYG_LDS_STACK)
// unit 4, the persistent kernels: their LDS holds the path states, the wide walks' stack stays in scratch.  Unit 5, wf_tick: one
// block runs the any-hit walk (one word per parked child, dscene.hpp wideStackLdsAny) and the closest-hit walk (two words,
// wideStackLds).  The two layouts are separate LDS arrays since round 3, so the hazard that put this unit's stack in scratch (a
// wave still in its shadow rays and a wave already in the next closest-hit walk overwriting each other) is gone; both arrays
// together are 48 KB per 256-thread block though, which would take wf_tick from its 2-3 blocks per CU to one, and the kernel is
// an option that measured slower than the tick loop either way: it keeps its stack in scratch.
#define ZYG_LDS_STACK 0
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

#if ZYG_VARIANTS
#define ZYG_IS_PERSISTENT(kid) (kid >= K_PERSISTENT_LAMBERT && kid <= K_PERSISTENT_SUBST_W)
#define ZYG_IS_REGENERATE(kid) (K_RGENERATE == kid)
#else
#define ZYG_IS_PERSISTENT(kid) false
#define ZYG_IS_REGENERATE(kid) false
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
    const dim3 regen_grid((blocks * 256u + kRegenerateThreads - 1u) / kRegenerateThreads), regen_block(kRegenerateThreads);
    switch (id) {
#define ZYG_X(unit, kid, kernel, args)                                                       \
    case kid:                                                                                \
        if constexpr (unit == UNIT) {                                                        \
            constexpr bool ts_u = K_TRACE_SHADE_PLAIN_FIRST_U == kid || K_TRACE_SHADE_PLAIN_U == kid ||              \
                                  (ZYG_TS_BLOCK_ALL && kid >= K_TRACE_SHADE_STATS_FIRST && kid <= K_TRACE_SHADE_PLAIN);      \
            constexpr bool ts_uw = K_TRACE_SHADE_PLAIN_FIRST_UW == kid || K_TRACE_SHADE_PLAIN_UW == kid;             \
            constexpr bool persistent = ZYG_IS_PERSISTENT(kid);                                                      \
            constexpr bool regenerate = ZYG_IS_REGENERATE(kid);                                                      \
            hipLaunchKernelGGL(kernel, ts_u ? ts_u_grid : ts_uw ? ts_uw_grid : regenerate ? regen_grid : grid,       \
                               ts_u ? ts_u_block : ts_uw ? ts_uw_block : persistent ? persistent_block : regenerate ? regen_block : block, 0, stream, ZYG_UNPAREN args); \
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

#if 7 == ZYG_TU
namespace zd {
// Ray-level hooks for the tests (zygpu_test_intersect / zygpu_test_visibility): rays of 8 floats (o.xyz, min_t, d.xyz, max_t)
// against the uploaded scene, in this unit's arithmetic (option `precise` picks the unit).  mode 0: the reference's walk (top-level tree lane by lane,
// BVH2); 1: the prop tables (sceneIntersectUniform) + BVH2; 2: the prop tables + the four-wide BLAS re-layout; 3: as the
// relaxed units walk: 2 without the tie rule between props and with the small meshes tested triangle by triangle.
__global__ void test_intersect(uint32_t n, const float* rays, uint32_t depth_surface, uint32_t mode, uint32_t* hits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + size_t(i) * 8;
    Ray          ray = Ray::init(mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7]);
    Hit          hit;
    Counters     cnt{0, 0, 0};
    bool         found;
    if (0 == mode) found = sceneIntersect<true, true>(c_scene, ray, depth_surface, hit, cnt);  // (the counting form: never the table, DScene::table_complete)
    else if (1 == mode) found = sceneIntersectUniform<true, false>(c_scene, ray, depth_surface, hit);
    else if (2 == mode) found = sceneIntersectUniform<true, true>(c_scene, ray, depth_surface, hit);
    else found = sceneIntersectUniform<false, true>(c_scene, ray, depth_surface, hit);  // no tie rule, small meshes triangle by triangle
    uint32_t* h = hits + size_t(i) * 8;
    for (int k = 0; k < 8; ++k) h[k] = 0;
    if (found) {
        h[0] = __float_as_uint(hit.t);
        h[1] = __float_as_uint(hit.u);
        h[2] = __float_as_uint(hit.v);
        h[3] = hit.primitive;
    }
    h[4] = hit.prop;
}
__global__ void test_visibility(uint32_t n, const float* rays, uint32_t mode, uint32_t* visible) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + size_t(i) * 8;
    Ray          ray = Ray::init(mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7]);
    Counters     cnt{0, 0, 0};
    visible[i] = (0 == mode   ? sceneVisibility<true, true>(c_scene, ray, cnt)  // (the counting form: the tree, not the table)
                  : 1 == mode ? sceneVisibilityUniform<false>(c_scene, ray)
                  : 2 == mode ? sceneVisibilityUniform<true>(c_scene, ray)
                              : sceneVisibilityUniform<true, true>(c_scene, ray))
                     ? 1u
                     : 0u;
}
}  // namespace zd
#endif

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

#if 7 == ZYG_TU
// kind 0: closest hit (8 words per ray), 1: visibility (one word per ray); device pointers
int ZYG_UNIT_SYM(test_rays)(int kind, unsigned n, const float* rays, unsigned depth_surface, unsigned mode, unsigned* out, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (0 == kind) hipLaunchKernelGGL(zd::test_intersect, dim3((n + 255) / 256), dim3(256), 0, s, n, rays, depth_surface, mode, out);
    else hipLaunchKernelGGL(zd::test_visibility, dim3((n + 255) / 256), dim3(256), 0, s, n, rays, mode, out);
    return 0;
}
#endif

}  // extern "C"
