// Every wavefront-kernel instantiation the library ships, with the translation unit that compiles it.
//
// The device code is built as several translation units in parallel (zyg_amd/build.py): kernels.hip is compiled once per
// ZYG_TU value and arithmetic mode, and instantiates exactly the entries of this table whose first column equals its
// ZYG_TU.  No relocatable device code: each unit has its own copy of the three __constant__ blocks (view, scene, Sobol
// table pointer), set through the unit's `set` entry.  zygpu.hip launches by id through the unit's `launch` entry.
//
//   X(unit, id, kernel instantiation, kernel arguments)       arguments: P = Params, arg = the launch's uint32_t
//
// Template parameters (wavefront.hpp):
//   wf_generate<POOL>   wf_intersect<STATS, POOL, EXTRA>   wf_shade<STATS, MESHL, HEAVY, POOL>
//   wf_light<MULTI, MASKS> + wf_scatter<.., PHASE 2>: the POOL scenes' complete wf_scatter in two kernels (wavefront.hpp scatterVertex)
//   wf_trace_shade<STATS, MESHL, FIRST, UTLAS>   wf_scatter<STATS, SUBST, MULTI, MESHL, POOL, MASKS, UTLAS>
//   wf_rgenerate (1024-thread blocks)   wf_rtrace<WIDE>   wf_rscatter<SUBST, WIDE> (the regenerating kernels; Params::regen_*)
//   wf_textures<POOL> (the vertices' procedural textures, dprocedural.hpp; unit ZYG_TEXTURES_UNIT)
//   wf_nee<FIRST, SUBST>   wf_shadow<STATS>   wf_shadow_u / wf_mesh_closest (persistent: launched with persistentGrid blocks)   wf_scatter_h<RESUME, SUBST>   wf_extend<SUBST, EXTRA>   wf_persistent<SUBST, MULTI, WIDE>   wf_tick<SUBST, MULTI, WIDE>
#pragma once

#define ZYG_NUM_UNITS 8

// the WIDE instantiations exist in the relaxed-arithmetic units only; in the precise ones their ids launch the BVH2 kernels
// (an alias must sit in the unit of the kernel it names: a kernel reads its own unit's __constant__ blocks)
#if ZYG_FAST
#define ZYG_WIDE(wide, plain) wide
#else
#define ZYG_WIDE(wide, plain) plain
#endif

// clang-format off
#define ZYG_KERNEL_TABLE_CORE(X)                                                            \
    X(0, K_SCATTER_STATS_POOL,   (wf_scatter<true, true, true, true, true>),          (P)) \
    X(1, K_SCATTER_MASKS_POOL,   (wf_scatter<false, true, true, true, true, true>),   (P)) \
    X(2, K_SCATTER_FULL_POOL,    (wf_scatter<false, true, true, true, true>),         (P)) \
    X(1, K_LIGHT_POOL,           (wf_light<true, false>),                             (P)) \
    X(0, K_LIGHT_POOL_RECT,      (wf_light<true, false, kRectangleLightKinds>),       (P)) \
    X(0, K_LIGHT_POOL_SKY,       (wf_light<true, false, kSkyLightKinds>),             (P)) \
    X(2, K_SCATTER_EVAL_POOL,    (wf_scatter<false, true, true, true, true, false, false, false, 2>), (P)) \
    X(2, K_SCATTER_EVAL_POOL_PLAIN, (wf_scatter<false, true, false, false, true, false, false, false, 2>), (P)) \
    X(2, K_SCATTER_PLAIN_POOL,   (wf_scatter<false, true, false, false, true>),       (P)) \
    X(3, K_SCATTER_STATS,        (wf_scatter<true, true, true, true, false>),         (P)) \
    X(7, K_SCATTER_MASKS,        (wf_scatter<false, true, true, true, false, true>),  (P)) \
    X(4, K_SCATTER_FULL,         (wf_scatter<false, true, true, true, false>),        (P)) \
    X(6, K_SCATTER_FULL_RECT,    (wf_scatter<false, true, true, true, false, false, false, false, 0, kRectangleLightKinds>), (P)) \
    X(4, K_SCATTER_MULTI,        (wf_scatter<false, true, true, false, false>),       (P)) \
    X(4, K_SCATTER_SUBST,        (wf_scatter<false, true, false, false, false>),      (P)) \
    X(4, K_SCATTER_LAMBERT,      (wf_scatter<false, false, false, false, false>),     (P)) \
    X(3, K_SCATTER_MULTI_U,      (wf_scatter<false, true, true, false, false, false, true>),   (P)) \
    X(6, K_SCATTER_SUBST_U,      (wf_scatter<false, true, false, false, false, false, true>),  (P)) \
    X(6, K_SCATTER_LAMBERT_U,    (wf_scatter<false, false, false, false, false, false, true>), (P)) \
    X(3, K_SCATTER_MULTI_UW,     ZYG_WIDE((wf_scatter<false, true, true, false, false, false, true, true>), (wf_scatter<false, true, true, false, false, false, true>)), (P)) \
    X(6, K_SCATTER_SUBST_UW,     ZYG_WIDE((wf_scatter<false, true, false, false, false, false, true, true>), (wf_scatter<false, true, false, false, false, false, true>)), (P)) \
    X(6, K_SCATTER_LAMBERT_UW,   ZYG_WIDE((wf_scatter<false, false, false, false, false, false, true, true>), (wf_scatter<false, false, false, false, false, false, true>)), (P)) \
    X(0, K_SHADE_STATS_LIGHT_POOL, (wf_shade<true, true, false, true>),               (P)) \
    X(0, K_SHADE_STATS_HEAVY_POOL, (wf_shade<true, true, true, true>),                (P)) \
    X(5, K_SHADE_PLAIN_LIGHT_POOL, (wf_shade<false, false, false, true>),             (P)) \
    X(5, K_SHADE_PLAIN_HEAVY_POOL, (wf_shade<false, false, true, true>),              (P)) \
    X(5, K_SHADE_MESHL_LIGHT_POOL, (wf_shade<false, true, false, true>),              (P)) \
    X(5, K_SHADE_MESHL_HEAVY_POOL, (wf_shade<false, true, true, true>),               (P)) \
    X(3, K_SHADE_MESHL_HEAVY_POOL_RECT, (wf_shade<false, true, true, true, kRectangleLightKinds>), (P)) \
    X(3, K_SHADE_MESHL_HEAVY_POOL_SKY,  (wf_shade<false, true, true, true, kSkyLightKinds>),       (P)) \
    X(1, K_SHADE_STATS_HEAVY,    (wf_shade<true, true, true, false>),                 (P)) \
    X(5, K_SHADE_MESHL_HEAVY,    (wf_shade<false, true, true, false>),                (P)) \
    X(5, K_SHADE_PLAIN_HEAVY,    (wf_shade<false, false, true, false>),               (P)) \
    X(6, K_SHADE_STATS_LIGHT,    (wf_shade<true, true, false, false>),                (P)) \
    X(6, K_SHADE_MESHL_LIGHT,    (wf_shade<false, true, false, false>),               (P)) \
    X(6, K_SHADE_PLAIN_LIGHT,    (wf_shade<false, false, false, false>),              (P)) \
    X(6, K_TRACE_SHADE_STATS_FIRST, (wf_trace_shade<true, true, true>),               (P, arg)) \
    X(6, K_TRACE_SHADE_MESHL_FIRST, (wf_trace_shade<false, true, true>),              (P, arg)) \
    X(6, K_TRACE_SHADE_PLAIN_FIRST, (wf_trace_shade<false, false, true>),             (P, arg)) \
    X(6, K_TRACE_SHADE_STATS,    (wf_trace_shade<true, true, false>),                 (P, arg)) \
    X(6, K_TRACE_SHADE_MESHL,    (wf_trace_shade<false, true, false>),                (P, arg)) \
    X(6, K_TRACE_SHADE_PLAIN,    (wf_trace_shade<false, false, false>),               (P, arg)) \
    X(6, K_TRACE_SHADE_PLAIN_FIRST_U, (wf_trace_shade<false, false, true, true>),     (P, arg)) \
    X(6, K_TRACE_SHADE_PLAIN_U,  (wf_trace_shade<false, false, false, true>),         (P, arg)) \
    X(6, K_TRACE_SHADE_PLAIN_FIRST_UW, ZYG_WIDE((wf_trace_shade<false, false, true, true, true>), (wf_trace_shade<false, false, true, true>)), (P, arg)) \
    X(6, K_TRACE_SHADE_PLAIN_UW, ZYG_WIDE((wf_trace_shade<false, false, false, true, true>), (wf_trace_shade<false, false, false, true>)), (P, arg)) \
    X(6, K_GENERATE_POOL,        (wf_generate<true>),                                 (P, arg)) \
    X(6, K_GENERATE,             (wf_generate<false>),                                (P, arg)) \
    X(7, K_INTERSECT_STATS_POOL, (wf_intersect<true, true, true>),                    (P)) \
    X(7, K_INTERSECT_PLAIN_POOL, (wf_intersect<false, true, false>),                  (P)) \
    X(7, K_INTERSECT_EXTRA_POOL, (wf_intersect<false, true, true>),                   (P)) \
    X(7, K_INTERSECT_EXTRA_POOL_WIDE, ZYG_WIDE((wf_intersect<false, true, true, true>), (wf_intersect<false, true, true>)), (P)) \
    X(7, K_POOL_RESET,           (wf_pool_reset<true>),                               (P)) \
    X(7, K_TEXTURES,             (wf_textures<false>),                                (P)) \
    X(7, K_TEXTURES_POOL,        (wf_textures<true>),                                 (P)) \
    X(7, K_TRACE_SHADE_POOL,     (wf_trace_shade_pool<false>),                        (P, arg)) \
    X(7, K_TRACE_SHADE_POOL_WIDE, ZYG_WIDE((wf_trace_shade_pool<true>), (wf_trace_shade_pool<false>)), (P, arg)) \
    X(6, K_TRACE_SHADE_POOL_FIRST, (wf_trace_shade_pool<false, true>),                (P, arg)) \
    X(6, K_TRACE_SHADE_POOL_FIRST_WIDE, ZYG_WIDE((wf_trace_shade_pool<true, true>), (wf_trace_shade_pool<false, true>)), (P, arg)) \
    X(7, K_INTERSECT_STATS,      (wf_intersect<true, false, true>),                   (P)) \
    X(7, K_INTERSECT_EXTRA,      (wf_intersect<false, false, true>),                  (P)) \
    X(7, K_INTERSECT_PLAIN,      (wf_intersect<false, false, false>),                 (P)) \
    X(7, K_NEE_FIRST_SUBST,      (wf_nee<true, true>),                                (P, arg)) \
    X(7, K_NEE_FIRST,            (wf_nee<true, false>),                               (P, arg)) \
    X(7, K_NEE_SUBST,            (wf_nee<false, true>),                               (P, arg)) \
    X(7, K_NEE,                  (wf_nee<false, false>),                              (P, arg)) \
    X(7, K_SHADOW_STATS,         (wf_shadow<true>),                                   (P, arg)) \
    X(7, K_SHADOW,               (wf_shadow<false>),                                  (P, arg)) \
    X(7, K_SHADOW_U,             ZYG_WIDE((wf_shadow_u<true>), (wf_shadow<false>)),         (P, arg)) \
    X(7, K_SCATTER_H,            ZYG_WIDE((wf_scatter_h<false, false>), (wf_shadow<false>)), (P, arg)) \
    X(7, K_SCATTER_H_SUBST,      ZYG_WIDE((wf_scatter_h<false, true>), (wf_shadow<false>)),  (P, arg)) \
    X(7, K_SCATTER_H_RESUME,     ZYG_WIDE((wf_scatter_h<true, false>), (wf_shadow<false>)),  (P, arg)) \
    X(7, K_SCATTER_H_RESUME_SUBST, ZYG_WIDE((wf_scatter_h<true, true>), (wf_shadow<false>)), (P, arg)) \
    X(7, K_SHADOW_U8,            ZYG_WIDE((wf_shadow_u8<true>), (wf_shadow<false>)),        (P, arg)) \
    X(7, K_MESH_CLOSEST8,        ZYG_WIDE((wf_mesh_closest8<true>), (wf_shadow<false>)),    (P, arg)) \
    X(7, K_EXTEND_EXTRA,         (wf_extend<true, true>),                             (P)) \
    X(7, K_EXTEND_SUBST,         (wf_extend<true, false>),                            (P)) \
    X(7, K_EXTEND,               (wf_extend<false, false>),                           (P))

// Kernels that lost their A/B on every BASELINE scene (wf_tick, wf_persistent, the regenerating wf_r* kernels, the
// four-wide wf_mesh_closest): not in the shipped library.  `python -m zyg_amd.build --variant all --units 4,5,6,7,-1 -DZYG_VARIANTS=1` builds
// variants/lib_all.so with them and with the options that select them (zygpu.hip), which is what their tests and tools load.
#ifndef ZYG_VARIANTS
#define ZYG_VARIANTS 0
#endif
#if ZYG_VARIANTS
#define ZYG_KERNEL_TABLE_VARIANTS(X)                                                        \
    X(5, K_TICK_LAMBERT,         (wf_tick<false, false, false>),                      (P)) \
    X(5, K_TICK_SUBST,           (wf_tick<true, false, false>),                       (P)) \
    X(5, K_TICK_LAMBERT_W,       ZYG_WIDE((wf_tick<false, false, true>), (wf_tick<false, false, false>)), (P)) \
    X(5, K_TICK_SUBST_W,         ZYG_WIDE((wf_tick<true, false, true>), (wf_tick<true, false, false>)), (P)) \
    X(4, K_PERSISTENT_LAMBERT,   (wf_persistent<false, false, false>),                (P, arg)) \
    X(4, K_PERSISTENT_SUBST,     (wf_persistent<true, false, false>),                 (P, arg)) \
    X(4, K_PERSISTENT_LAMBERT_W, ZYG_WIDE((wf_persistent<false, false, true>), (wf_persistent<false, false, false>)), (P, arg)) \
    X(4, K_PERSISTENT_SUBST_W,   ZYG_WIDE((wf_persistent<true, false, true>), (wf_persistent<true, false, false>)), (P, arg)) \
    X(7, K_MESH_CLOSEST,         ZYG_WIDE((wf_mesh_closest<true>), (wf_shadow<false>)),     (P, arg)) \
    X(6, K_RGENERATE,            (wf_rgenerate<false>),                               (P)) \
    X(6, K_RTRACE,               (wf_rtrace<false>),                                  (P)) \
    X(6, K_RSCATTER,             (wf_rscatter<false, false>),                         (P)) \
    X(6, K_RSCATTER_SUBST,       (wf_rscatter<true, false>),                          (P))
#else
#define ZYG_KERNEL_TABLE_VARIANTS(X)
#endif
#define ZYG_KERNEL_TABLE(X) ZYG_KERNEL_TABLE_CORE(X) ZYG_KERNEL_TABLE_VARIANTS(X)
// clang-format on

enum ZygKernelId : unsigned {
#define ZYG_X(unit, id, kernel, args) id,
    ZYG_KERNEL_TABLE(ZYG_X)
#undef ZYG_X
        K_NUM_KERNELS
};

// the per-unit entries (C linkage, one pair per unit and arithmetic mode: zyg_unit<k>_launch / zyg_unit<k>f_launch)
enum ZygUnitConstant : int { ZYG_CONST_VIEW = 0, ZYG_CONST_SCENE = 1, ZYG_CONST_SOBOL_LUT = 2 };
