// The wavefront stages (gfx950, wave64).  One batch = B iterations x every sample pixel of the row range, one path
// per (pixel, iteration), path slot == sample index.  Per tick (= one path vertex) the stages run over device-side
// queues of slot indices:
//   wf_intersect : closest hit of the extension rays            Scene.intersect   (reference prop_tree.zig:56-116)
//   wf_shade     : emission at the hit + unoccluding lights, MIS weight, termination, Russian roulette, closure set-up;
//                  writes the compacted hit record               connectLight + Vertex.sample (pathtracer_mis.zig:45-92, 280-341)
//   wf_nee(r)    : round r of next-event estimation: light-tree picks (r = 0), Rectangle sampling, shadow ray + the
//                  pre-weighted contribution into the pool           sampleLights / evaluateLight (pathtracer_mis.zig:174-278)
//   wf_shadow    : any-hit test of the round's shadow rays       Scene.visibility  (prop_tree.zig:185-240)
//   wf_extend    : last pick's tail, bxdf sample, path state update, next extension ray (pathtracer_mis.zig:123-166)
// Every launch is one thread per queue entry (no grid-stride loop: hoisted loop invariants cost ~60 VGPRs here); the
// queue lengths live on the device, so a launch covers the upper bound and surplus waves retire at once.
//
// POOL instantiations (scenes with Glass): PathtracerMIS splits a path at dielectric interfaces into a reflected and a
// refracted one, up to 4 per camera sample (Vertex.Pool, vertex.zig:214-310).  The vertices of one level are consumed in
// order and share the sample's samplers, so a tick runs 4 rounds of intersect / shade / scatter: round k handles vertex k of
// every sample's current level.  Records are then indexed by vertex = (half * 4 + k) * batch + slot (two halves: current
// and next level), the samplers, the IValue accumulators and the alpha sum by slot.
#pragma once

#include "dmaterial.hpp"

namespace zd {

// meta word: [0:2] sobol dimension, [3:7] depth_surface, [8:12] vertex.depth, [13:18] state flags (vertex.zig:19-28)
enum : uint32_t {
    ST_PRIMARY = 1u << 0,
    ST_TRANSPARENT = 1u << 1,
    ST_SINGULAR = 1u << 2,
    ST_SPECULAR = 1u << 3,
    ST_TRANSLUCENT = 1u << 4,
    ST_STARTED_SPECULAR = 1u << 5,
    ST_FROM_CATCHER = 1u << 6,       // vertex.state.from_shadow_catcher (vertex.zig:26)
    ST_CATCHER_IN_CAMERA = 1u << 7   // vertex.state.shadow_catcher_in_camera (vertex.zig:27)
};

// The records every stage touches live interleaved, one 256 B block (two 128 B lines) per path slot, instead of in one array
// each: slot == sample index, so the living slots thin out with depth (Cornell: ~80 % survive a bounce), and a 16 B record
// of a separate array then costs a whole 64 B sector / 128 B line of which a quarter or less is used.  Measured with
// separate arrays (PMC FETCH_SIZE, profiles/): wf_shade's first pass fetched 320 B per vertex for ~110 B of records.
// With the block, a path's records share their lines at any density.  ZYG_POOL_STRIDE 0 restores the separate arrays.
#ifndef ZYG_POOL_STRIDE
#define ZYG_POOL_STRIDE 256
#endif
// hit, hit_prop, ucount: 32 = a block of their own, one 32 B sector per path, written by wf_intersect and read by wf_shade
// (inside the 256 B block, ZYG_HIT_STRIDE 0, wf_intersect's partial-line writes cost it 8 %; measured 414 vs 427 Msamples/s)
#define ZYG_HIT_STRIDE 32
#if ZYG_POOL_STRIDE
// A record of the pool behind a proxy, so that the access is in one place.  (Measured and dropped: the non-temporal hint on
// these loads and stores, meant to keep the scene data in the vector L1: 556 -> 266 Msamples/s, the hint also costs the records
// their L2 hits between the stage that writes them and the stage that reads them.)
template <typename T>
struct RecRef {
    T* p;
    ZD operator T() const { return *p; }
    ZD const RecRef& operator=(const T& v) const {
        *p = v;
        return *this;
    }
    ZD const RecRef& operator=(const RecRef& o) const { return *this = T(o); }
};
// `mask` / `local`: the wavefront kernels index the records by path slot (mask = ~0, local = 0); wf_persistent keeps one
// path's block per lane in LDS and reaches it whatever the slot (mask = 0, local = the lane's index in the block).
// `stride`: the bytes between two slots' records: STRIDE (the interleaved block) in the pool of the tick loop; the regenerating
// kernels (wf_rtrace / wf_rscatter), whose slots are all alive, keep every record in an array of its own (16 B, the hit block 32 B)
// that a wave reads and writes 1 KB at a time.
template <typename T, int STRIDE = ZYG_POOL_STRIDE>
struct Rec {
    static constexpr uint32_t kBlockStride = STRIDE;
    char*    base;  // block base + the member's offset (RecOffsets)
    uint32_t mask, local;
    uint32_t stride;
    ZD RecRef<T> operator[](size_t i) const {
        return RecRef<T>{reinterpret_cast<T*>(base + size_t((uint32_t(i) & mask) + local) * size_t(stride))};
    }
    ZD T get(size_t i) const { return T((*this)[i]); }
};
template <typename T>
using HitRec = Rec<T, ZYG_HIT_STRIDE ? ZYG_HIT_STRIDE : ZYG_POOL_STRIDE>;
#else
template <typename T>
using Rec = T*;
template <typename T>
using HitRec = T*;
#endif
// line A: what wf_intersect and wf_shade's first pass touch; line B: the rest of the hit record and the previous vertex
enum RecOffsets : uint32_t {
    kOffQ0 = 0, kOffQ1 = 16, kOffPcg = 32, kOffRayO = 48, kOffRayD = 64, kOffThr = 80,
#ifdef ZYG_ACC_SEPARATE
    kOffRecP = 112,
#else
    // the IValue accumulator wf_scatter adds to at every bounce (indirect) sits in the line it rewrites anyway (as a separate
    // array each add was a read-modify-write of one more sparse sector); the two others share a sector nobody else touches
    kOffIndirect = 96, kOffRecP = 112, kOffEmission = 224, kOffDirect = 240,
#endif
    kOffRecGn = 128, kOffRecN = 144, kOffRecT = 160, kOffRecB = 176, kOffVorigin = 192, kOffVgeoN = 208,
    kOffHit = 96, kOffHitProp = 40, kOffUcount = 44  // with ZYG_HIT_STRIDE 0 only
};
#if ZYG_POOL_STRIDE && !defined(ZYG_ACC_SEPARATE)
template <typename T>
using AccRec = Rec<T>;
#else
template <typename T>
using AccRec = T*;
#endif

struct Pool {  // 16 B records, indexed by path slot
    Rec<uint4>    q0;       // sobol run_seed, meta, sobol buffer[4] bits, NEE state
    Rec<float4>   q1;       // sobol buffer[0..3]
    Rec<uint2>    pcg;      // PCG state
    Rec<float4>   ray_o;    // extension ray origin.xyz, max_t
    Rec<float4>   ray_d;    // extension ray direction.xyz, vertex.bxdf_pdf
    Rec<float4>   thr;      // throughput.xyz
    Rec<float4>   vorigin;  // vertex.origin
    Rec<float4>   vgeo_n;   // vertex.geo_n
    AccRec<float4> emission; // IValue (helper.zig:6-19)
    AccRec<float4> direct;
    AccRec<float4> indirect;
    HitRec<uint4>    hit;      // t, u, v (bits), primitive
    HitRec<uint32_t> hit_prop;
    HitRec<float>    hit_sr;   // rs.stochastic_r of Vertex.sample (vertex.zig:165), drawn by wf_shade, used by the closure set-up
    HitRec<uint32_t> ucount;   // unoccluding lights the extension ray crosses in front of its hit (PropBvh.emission's candidates)
    uint4*    ucand;    //   [candidate][slot]: prop, t, u, v (bits), in traversal order
    uint32_t* ucand_prim; //                    triangle (mesh lights)
    Rec<float4>   rec_p;    // compacted hit record: p.xyz, material index (bits)
    Rec<float4>   rec_gn;   //   geometric normal after the two-sided flip (vertex.zig:155-161), w = 1 if flipped
    Rec<float4>   rec_n;    //   shading normal (flipped likewise)
    Rec<float4>   rec_t;    //   tangent
    Rec<float4>   rec_b;    //   bitangent
    uint32_t* time;     // per camera sample: Probe.time as ticks since the frame's start (written when the scene has animated props)
    uint2*    picks;    // light picks 1.. of the current vertex, [pick][slot]: light, pdf bits (pick 0 never leaves registers)
    float4*   sh_o;     // shadow ray of the running pick: origin.xyz, max_t
    float4*   sh_d;     //   direction.xyz
    float4*   contrib;  //   its pre-weighted contribution weight * radiance * bsdf
    float4*   nee;      // sum of the unoccluded contributions of this vertex
    uint32_t* vis;      // wf_shadow's verdict on the running pick
    // POOL only.  The POOL pool's layout (zygpu.hip ensurePool): a vertex is two lines of its own 256 B block, a camera sample one
    // 128 B line: wf_scatter<POOL> reads its vertex' two lines and the sample's, writes a child's first line and the sample's.  (Until
    // round 4 these seven were arrays of their own beside the non-POOL block: ~20 lines per scattered vertex, 3.6 KB of HBM traffic.)
    //   vertex, line A: ray_o 0, ray_d 16, thr 32, vorigin 48, vgeo_n 64, med0 80, med1 96, vmeta 112, hit_sr 116, rec_ior 120
    //           line B: rec_p 128, rec_gn 144, rec_n 160, rec_t 176, rec_b 192, hit 208, hit_prop 224, ucount 228, rec_uv 232
    //   sample:         q0 0, q1 16, pcg 32, pstate 40, transp 44, emission 48, direct 64, indirect 80
    Rec<uint32_t> vmeta;    // per vertex: depth_surface, vertex.depth, state flags (the bits of q0.y above the sobol dimension)
    Rec<uint4>    med0;     // per vertex: medium stack (prop/medium.zig:30-153): index, prop0, part0, prop1
    Rec<uint4>    med1;     //             part1, prop2, part2
    Rec<float2>   rec_ior;  // per vertex: rs.ior (Vertex.iorOutside), 1 if the material has lower priority than the stack
    Rec<float2>   rec_uv;   // per vertex: the fragment's uv, for the textures of an extended Substitute
    Rec<uint32_t> pstate;   // per slot: [0:7] vertices in flight, [8:15] children created for the next level
    Rec<float>    transp;   // per slot: Pool.transparency[3], the alpha sum over terminated vertices
    // scenes with a shadow catcher only (else null), PathtracerMIS: per vertex what a catcher on the path has seen (vertex.zig:57-59),
    // per slot Pool.transparency's colour (the backplate light a catcher lets through, vertex.zig:252-266)
    float4*   sc_occ;
    float4*   sc_unocc;
    float4*   sc_emis;
    float4*   transp_rgb;
};

constexpr uint32_t kPoolVertices = 4;  // Vertex.Pool.NumVertices

struct Queues {
    uint32_t* ext[2];  // extension rays, ping-pong per tick
    uint32_t* ext_pool[2][kPoolVertices];  // POOL: one queue per round, [1] = the extra queues of rounds 1..3
    uint32_t* alive;   // paths that survived wf_shade in this tick
    uint32_t* shadow;  // paths with a shadow ray in flight (one NEE round)
    uint32_t* heavy;   // extension rays whose vertex has emission to evaluate (wf_shade's second pass)
};

// Queue lengths live on the device: the front and the back end's counts in separate 64 B lines (their atomics do not
// share an address).  Measured on configs[1]: one atomic per wave on one word 248 Msamples/s, one per block 303; spreading
// the appends over 64 run counters cost more in the consumers than the remaining contention (278), so there is one run.
constexpr uint32_t kCountStride = 16;
// The alive queue sorted: sub-queue k holds the paths whose hit prop is k (mod kAliveKeys), each in its own array of
// queue_capacity entries and consumed in whole 256-thread blocks, so that the lanes of a wf_scatter wave stand on one surface:
// one fragment kind and one material per wave, the light sample succeeds or fails for the whole wave, the shadow rays start
// side by side.  Params::alive_keys = 1: everything in sub-queue 0 (the plain queue).
constexpr uint32_t kAliveKeys = 16;
struct QueueCounts {
    uint32_t n[2 * kCountStride];  // [0] entries appended at the front, [kCountStride] at the back
};

struct KernelCounters {
    unsigned long long nodes, tris, instances, rays;
};

struct Control {  // device-resident
    unsigned long long next_sample;  // wf_persistent: the next camera sample of the batch to hand out
    QueueCounts        ext_count[2];
    QueueCounts        ext_pool_count[2][kPoolVertices];  // POOL
    QueueCounts        alive_count[2][kAliveKeys];  // [parity of the tick][sub-queue]: wf_trace_shade fills this tick's and resets the next one's
    QueueCounts        shadow_count[2];  // ping-pong per NEE round
    QueueCounts        heavy_count[2];
    uint32_t           bound[2];  // [parity of the tick]: length of the tick's extension queue = upper bound of every later
                                  // queue of that tick; surplus blocks leave on this one scalar load
    uint32_t           error;
    uint32_t           shadow_fetch;  // wf_shadow_u: the next queue entry to hand out; the round's wf_nee zeroes it
    uint32_t           mesh_fetch;    // wf_mesh_closest: likewise; the tick's wf_trace_shade zeroes it for the next tick
    uint32_t           regen_live;    // wf_rscatter (asked to report): some slot still carries a path
    KernelCounters     k[3];  // 0 wf_intersect, 1 wf_shade, 2 wf_shadow
    unsigned long long sections[32];  // -DZYG_PROFILE_SECTIONS builds: shader cycles per code section, summed over waves
};

struct View {
    ZygpuViewDesc v;
    int32_t       y_begin, y_end;  // sample rows incl. filter padding
    int32_t       x_begin, x_end;  // sample columns incl. filter padding
    uint32_t      frame, iteration0, batch_iterations;
    uint32_t      npix;  // samples per iteration in this batch
    // One-shot rendering (zygpu_render_frame = Worker.render with num_samples > 1, worker.zig:141-165): the pixel's PCG stream
    // is started once, at `oneshot_base`, and runs on across its samples; a batch is one sample per pixel and
    // iteration0 = oneshot_base + the sample's number.  0 = progressive: every (pixel, iteration) starts its own stream.
    uint32_t      one_shot, oneshot_base;
};

struct Params {
    Pool     pool;
    Queues   q;
    Control* ctl;
    uint2*   pixel_pcg;    // one-shot rendering: per sample pixel the PCG state its last sample left (else null)
    float4*  stage_color;  // per sample: clamped composed colour, alpha
    float2*  stage_uv;     // per sample: pixel_uv
    float4*  film;
    float*   film_alpha;  // Transparent sensor only (else null): sum of weight * alpha per pixel
    // AOVs (c_view.v.aov_slots != 0): per sample the worker's aov.Value, classes Albedo .. Roughness in planes of `aov_plane`
    // samples; per pixel the Pack4f plane of the class film_accumulate_aov is launched for
    float4*  stage_aov;
    size_t   aov_plane;
    float4*  aov_film;
    // the queue of the paths that survived shading: wf_trace_shade / wf_shade append to alive_out, wf_scatter (wf_nee, wf_extend)
    // read alive_in.  Both are q.alive when the stages of a tick run apart; wf_tick reads the previous tick's and writes this
    // tick's (the two ext arrays, which it does not need).
    uint32_t* alive_in;
    uint32_t* alive_out;
    uint32_t alive_keys;  // sub-queues of the alive queue in use: 1, or kAliveKeys (sorted by hit prop)
    // the regenerating kernels (option "regenerate"): per pool slot the camera sample (index in the batch) its path belongs to, kNull
    // when the slot has none; null in the tick loop, where a path's slot is its sample's index
    uint32_t* regen_sample;
    uint32_t  regen_slots, regen_samples;  // slots of the pool, camera samples of the batch
    uint32_t  regen_report;                // 1: the launch after which the host reads Control::regen_live
    uint32_t parity;  // current extension-queue side
    uint32_t queue_capacity;  // entries every queue array holds
    uint32_t packed_acc;      // 1: the paths keep emission / direct in their first line (kPackedAccumulators; set for the non-POOL kernels)
    // 1: the previous vertex (vertex.origin, vertex.geo_n, its roughness) travels in words of the path's first line that are dead
    // while it is needed (PrevVertex below) instead of in two records of the second line; the non-POOL kernels of the tick loop in
    // scenes without opacity masks
    uint32_t packed_prev;
    // The hit records of the vertices that go on to wf_scatter, indexed by their position in the tick's alive queue instead of by path
    // slot (else null): with depth the surviving slots thin out, and a 32 B hit block of the slot-indexed array then costs its whole
    // 128 B line, as much as the path's state; at the queue's positions the records are dense: a wave of wf_scatter reads 2 KB in a
    // row.  Two uint4 per entry: (t, u, v, primitive), (prop, unoccluding count, stochastic_r, -).  wf_trace_shade / wf_shade write
    // them behind their append to the queue; the vertices deferred to wf_shade's second pass keep the slot-indexed record in between.
    uint4* hit_q;
    // The split next-event estimation of the POOL scenes' complete kernels (wf_light + wf_scatter<.., PHASE 2>, below): per camera
    // sample kMaxNeeStored light samples that reached their light (three float4 each: wi + light pdf; radiance + pick; transmittance +
    // the pick's shadow-catcher flag) and their count (kNeeOverflow: the vertex took more and goes through the unsplit kernel)
    float4*   nee_rec;
    uint32_t* nee_count;
    uint32_t  scatter_source;  // wf_scatter reads 0: the alive queue, 1: the overflow queue (Queues::shadow, shadow_count[0]) of wf_light
    uint32_t round;   // POOL: which vertex of the level this launch handles
    uint32_t mesh_pass;  // 1: wf_mesh_closest has traced the back entries of the extension queue; wf_trace_shade reads their hits; 2: wf_mesh_closest8 has walked the large meshes for them, wf_trace_shade tests the rest of the table
    uint32_t alive_key_mask;  // wf_scatter_h: bit k = the launch takes sub-queue k of the sorted alive queue
    uint32_t lazy_head;  // 1 (option "lazy_head", with hit_q): the first shading pass leaves a vertex' sampler and throughput to wf_scatter (shadeVertex)
    // The values of the vertices' procedural textures (dprocedural.hpp; null: the scene has none): a block of kTexPlanes floats (64 B, half
    // a line: one fill, one write) per path slot (POOL: per vertex), indexed like the pool's records.  wf_textures writes the whole block
    // between the shading passes and the kernels that build the closure, which read it where they would have fetched a texel
    // (FragmentExtra::tex, loadSurface).  (One plane per value, the round's first layout, cost a sector per value and vertex: the living
    // slots are sparse.)
    float*   tex_values;
    // POOL scenes: the alive queue has two ends.  A vertex whose closure cannot be evaluated whatever its fragment (smooth Glass without a
    // roughness map or regularisation: no light sample is taken or evaluated there, pathtracer_mis.zig:185-187) is appended from the back;
    // wf_light reads the front entries only (alive_ends 2), every other consumer both ends (1).  0: one end (the other scenes)
    uint32_t alive_ends;
};

// one copy per translation unit (kernel_table.hpp): the units are built without relocatable device code
// (names and visibility: dsampler.hpp ZYG_CONSTANT_BLOCK)
#define c_view ZYG_BLOCK_NAME(c_view)
#define c_scene ZYG_BLOCK_NAME(c_scene)
ZYG_CONSTANT_BLOCK View   c_view;
ZYG_CONSTANT_BLOCK DScene c_scene;  // the compiled scene's device arrays: read on demand through the scalar cache

// occupancy targets (waves per SIMD) the register allocator is held to, per kernel; see DESIGN.md "Occupancy"
#ifndef ZYG_SHADE_WAVES
#define ZYG_SHADE_WAVES 3
#endif
#ifndef ZYG_SHADE_BIG_WAVES  // the second pass with mesh / sphere / infinite / image light code: 2 waves instead of ~250 spilled registers
#define ZYG_SHADE_BIG_WAVES 2
#endif
#ifndef ZYG_SHADE_LIGHT_WAVES
#define ZYG_SHADE_LIGHT_WAVES 5
#endif
#ifndef ZYG_TRACE_SHADE_WAVES
#define ZYG_TRACE_SHADE_WAVES 4
#endif
#ifndef ZYG_SCATTER_WAVES
#define ZYG_SCATTER_WAVES 3
#endif
#ifndef ZYG_SCATTER_BIG_WAVES  // instantiations with Substitute / multi-sample / mesh-light code: 2 waves with ~20 spilled
#define ZYG_SCATTER_BIG_WAVES 2  // registers beat 3 waves with 120-250 (measured on configs[2]: 174 -> 158 ms per 3 steps)
#endif
#ifndef ZYG_NEE_WAVES
#define ZYG_NEE_WAVES 3
#endif
#ifndef ZYG_EXTEND_WAVES
#define ZYG_EXTEND_WAVES 3
#endif
#define ZYG_WAVES(n) __attribute__((amdgpu_waves_per_eu(n, n)))

// ---------------------------------------------------------------- helpers

// Section timer of the profiling build (tools/sections.sh): s_memtime deltas per wave, summed into Control::sections.
#ifdef ZYG_PROFILE_SECTIONS
struct SectionTimer {
    unsigned long long last, acc[16];
    bool               lanes = false;  // sections 0-7 only: acc[8 + id] = the section's cycles x the lanes that reached its mark
    ZD void start(bool with_lanes = false) {
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        lanes = with_lanes;
        last = clock64();
    }
    ZD void mark(int id) {
        const unsigned long long t = clock64();
        acc[id] += t - last;
        if (lanes) acc[8 + id] += (t - last) * __popcll(__ballot(1));
        last = t;
    }
    ZD void flush(Control* ctl, int base) {
        // every lane of a wave holds the same sums for the sections it passed; take the per-wave maximum
        for (int i = 0; i < 16; ++i) {
            unsigned long long v = acc[i];
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long w = __shfl_xor(v, o);
                v = w > v ? w : v;
            }
            if (0 == __lane_id() && v) atomicAdd(&ctl->sections[base + i], v);
        }
    }
};
#else
struct SectionTimer {
    ZD void start(bool = false) {}
    ZD void mark(int) {}
    ZD void flush(Control*, int) {}
};
#endif

// Append of the active lanes' values to a queue: one atomic per 256-thread block and queue end (every wave of the block that
// has not returned must call it; waves count their lanes by ballot, the block totals are claimed by one atomicAdd each,
// every wave writes its part).  Queue order is free: all results are addressed by slot.
// A queue has two ends: `back` entries grow down from the last element.  The extension queue uses that to keep the rays
// that will enter a mesh's bounding box (deep BLAS walks) apart from the rest, so that the lanes of a wave have similar
// work: measured lane utilisation of wf_intersect on the 1 M-triangle scene was 11 % with mixed waves.
// Threads of a wf_trace_shade block (kernels.hip launches it so).  The instantiations that walk the prop tables and no wide BLAS
// (configs[1]) run 512-thread blocks: half the appends (atomics), and the alive queue comes out in longer runs of neighbouring
// paths: wf_trace_shade -1.1 %, wf_scatter -1 % against 256 in one call; 128: +6 %, 64: +90 %.  The wide-BLAS ones keep 256 (their
// LDS stack is sized for it and their waves' walks differ more in length); for the others (-DZYG_TS_BLOCK_ALL=1) 512 measures the same
// on the outdoor scene.  End of round 3: with the geometry tests in their own frames the kernel issues half the instructions, and what
// a block of eight waves costs at the append's barriers outweighs the halved atomics: 1024 / 512 / 256 / 128 threads 964 / 989-997 /
// 1009-1022 / 901 Msamples/s in one call each (wf_trace_shade 112.3 / 102.9-104.4 / 95.9-97.6 / 127.4 ms per 8 steps): 256 again.
#ifndef ZYG_TS_BLOCK
#define ZYG_TS_BLOCK 256
#endif
#ifndef ZYG_TS_BLOCK_WIDE
#define ZYG_TS_BLOCK_WIDE 256  // with -DZYG_LDS_STACK_THREADS=<the same> (dscene.hpp)
#endif
template <bool UTLAS, bool WIDE>
constexpr uint32_t traceShadeBlock() {
#if ZYG_TS_BLOCK_ALL
    return !WIDE ? uint32_t(ZYG_TS_BLOCK) : uint32_t(ZYG_TS_BLOCK_WIDE);
#else
    return (UTLAS && !WIDE) ? uint32_t(ZYG_TS_BLOCK) : WIDE ? uint32_t(ZYG_TS_BLOCK_WIDE) : 256u;
#endif
}

template <int WAVES = 4>
ZD void queuePush(uint32_t* queue, QueueCounts* counts, uint32_t capacity, bool active, uint32_t value, bool back = false,
                  uint32_t* position = nullptr) {
    __shared__ uint32_t wave_count[WAVES][2];
    __shared__ uint32_t block_base[2];
    const bool               front = active && !back, rear = active && back;
    const unsigned long long fmask = __ballot(front), rmask = __ballot(rear);
    const uint32_t           lane = __lane_id(), wave = threadIdx.x >> 6;
    if (0 == lane) {
        wave_count[wave][0] = uint32_t(__popcll(fmask));
        wave_count[wave][1] = uint32_t(__popcll(rmask));
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint32_t e = threadIdx.x;
        uint32_t       total = 0;
        for (int w = 0; w < WAVES; ++w) total += wave_count[w][e];
        block_base[e] = total ? atomicAdd(&counts->n[e * kCountStride], total) : 0u;
    }
    __syncthreads();
    const uint32_t e = rear ? 1u : 0u;
    uint32_t       index = block_base[e];
    for (uint32_t w = 0; w < wave; ++w) index += wave_count[w][e];
    index += uint32_t(__popcll((rear ? rmask : fmask) & ((1ull << lane) - 1ull)));
    if (active) queue[rear ? capacity - 1u - index : index] = value;
    if (position) *position = rear ? capacity - 1u - index : index;
    // no trailing barrier: a kernel that pushes twice puts its own barrier between the calls
}

// Two appends with one pair of barriers: the lanes with `a` go to the front of queue A, those with `b` to the front of queue B.
template <int WAVES = 4>
ZD void queuePushTwo(uint32_t* queue_a, QueueCounts* counts_a, bool a, uint32_t* queue_b, QueueCounts* counts_b, bool b, uint32_t value,
                     uint32_t* position = nullptr) {
    __shared__ uint32_t wave_count[WAVES][2];
    __shared__ uint32_t block_base[2];
    const unsigned long long amask = __ballot(a), bmask = __ballot(b);
    const uint32_t           lane = __lane_id(), wave = threadIdx.x >> 6;
    if (0 == lane) {
        wave_count[wave][0] = uint32_t(__popcll(amask));
        wave_count[wave][1] = uint32_t(__popcll(bmask));
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint32_t e = threadIdx.x;
        uint32_t       total = 0;
        for (int w = 0; w < WAVES; ++w) total += wave_count[w][e];
        block_base[e] = total ? atomicAdd(&(e ? counts_b : counts_a)->n[0], total) : 0u;
    }
    __syncthreads();
    const uint32_t e = b ? 1u : 0u;
    uint32_t       index = block_base[e];
    for (uint32_t w = 0; w < wave; ++w) index += wave_count[w][e];
    index += uint32_t(__popcll((b ? bmask : amask) & ((1ull << lane) - 1ull)));
    if (a || b) (b ? queue_b : queue_a)[index] = value;
    if (position) *position = index;
}

// Consumer side: the queue's total length and, for entry i (front entries first, then the back ones), its position.
ZD uint32_t queueOpen(const QueueCounts* counts, uint32_t capacity, uint32_t i, uint32_t& position) {
    const uint32_t front = counts->n[0], rear = counts->n[kCountStride];
    position = i < front ? i : capacity - 1u - (i - front);
    return front + rear;
}

ZD void queueReset(QueueCounts* counts) {  // by one wave
    if (__lane_id() < 2) counts->n[__lane_id() * kCountStride] = 0;
}
ZD void aliveReset(QueueCounts (&counts)[kAliveKeys]) {  // by one wave
    if (__lane_id() < kAliveKeys) counts[__lane_id()].n[0] = 0;
    if (kAliveKeys == __lane_id()) counts[0].n[kCountStride] = 0;  // (the back end of the POOL scenes' two-ended queue)
}

// Consumer side of the (sorted) alive queue for 256-thread blocks: thread i's entry.  Sub-queue k's entries come first to last in
// whole blocks (its length rounded up to 256), so a block, and with it a wave, reads one sub-queue.  Returns whether the
// thread belongs to a block that has entries at all (whole blocks take part in the queue appends that follow); `valid`: it has an
// entry, at alive_in[at].
ZD bool aliveOpen(const Params& P, uint32_t parity, uint32_t i, uint32_t& at, bool& valid, uint32_t* key = nullptr) {
    if (P.alive_ends) {  // the POOL scenes' two-ended queue: front entries first, then (unless the launch is wf_light's) the back ones
        const uint32_t front = P.ctl->alive_count[parity][0].n[0], rear = 2u == P.alive_ends ? 0u : P.ctl->alive_count[parity][0].n[kCountStride];
        valid = i < front + rear;
        at = i < front ? i : P.queue_capacity - 1u - (i - front);
        if (key) *key = i < front ? 0u : 1u;
        return (i & ~255u) < front + rear;
    }
    const uint32_t block_first = i & ~255u;  // (uniform over the block)
    uint32_t       begin = 0;
    valid = false;
    at = 0;
    for (uint32_t k = 0; k < P.alive_keys; ++k) {
        const uint32_t n = P.ctl->alive_count[parity][k].n[0];
        const uint32_t end = begin + ((n + 255u) & ~255u);
        if (block_first < end) {
            at = k * P.queue_capacity + (i - begin);
            valid = i - begin < n;
            if (key) *key = k;
            return true;
        }
        begin = end;
    }
    return false;
}
// the surplus blocks of a launch leave on this: the tick's extension-queue length bounds the alive entries, each sub-queue
// rounds its part up to a block
ZD uint32_t aliveBound(const Params& P, uint32_t parity) { return P.ctl->bound[parity] + (P.alive_keys > 1 ? P.alive_keys * 256u : 0u); }

// Producer side: the lanes with `defer` go to the heavy queue, those with `alive` to sub-queue `key` of the alive queue; one pair
// of barriers, one atomic per block and non-empty queue.
template <int WAVES>
ZD void alivePushKeyed(const Params& P, uint32_t cur, bool defer, bool alive, uint32_t key, uint32_t value) {
    constexpr uint32_t kHeavy = kAliveKeys;  // the heavy queue's index in the tables below
    __shared__ uint32_t wave_count[WAVES][kAliveKeys + 1];
    __shared__ uint32_t block_base[kAliveKeys + 1];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t mine = defer ? kHeavy : alive ? key : ~0u;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t rank = 0;
    for (uint32_t k = 0; k <= kAliveKeys; ++k) {
        const unsigned long long m = __ballot(mine == k);
        if (0 == lane) wave_count[wave][k] = uint32_t(__popcll(m));
        if (mine == k) rank = uint32_t(__popcll(m & below));
    }
    __syncthreads();
    if (threadIdx.x <= kAliveKeys) {
        const uint32_t k = threadIdx.x;
        uint32_t       total = 0;
        for (int w = 0; w < WAVES; ++w) total += wave_count[w][k];
        QueueCounts* counts = kHeavy == k ? &P.ctl->heavy_count[cur] : &P.ctl->alive_count[cur][k];
        block_base[k] = total ? atomicAdd(&counts->n[0], total) : 0u;
    }
    __syncthreads();
    if (~0u != mine) {
        uint32_t index = block_base[mine] + rank;
        for (uint32_t w = 0; w < wave; ++w) index += wave_count[w][mine];
        if (kHeavy == mine) P.q.heavy[index] = value;
        else P.alive_out[size_t(mine) * P.queue_capacity + index] = value;
    }
}

// the camera sample (index in the batch: pixel and iteration, the staging buffer's entry) of the path in pool slot `slot`
ZD uint32_t sampleOf(const Params& P, uint32_t slot) { return P.regen_sample ? P.regen_sample[slot] : slot; }

ZD void slotPixel(uint32_t slot, int& x, int& y, uint32_t& iteration) {
    const uint32_t it = slot / c_view.npix, pix = slot - it * c_view.npix;
    const uint32_t w = uint32_t(c_view.x_end - c_view.x_begin);
    y = c_view.y_begin + int(pix / w);
    x = c_view.x_begin + int(pix % w);
    iteration = c_view.iteration0 + it;
}

ZD uint32_t batchPaths() { return c_view.npix * c_view.batch_iterations; }

// POOL addressing: a queue entry is a vertex index; the camera sample it belongs to; the stride between the planes of
// per-vertex multi-entry arrays (candidates)
template <bool POOL> ZD uint32_t sampleSlot(uint32_t vertex) { return POOL ? vertex % batchPaths() : vertex; }
template <bool POOL> ZD size_t   vertexPlane() { return POOL ? size_t(2 * kPoolVertices) * batchPaths() : size_t(batchPaths()); }
template <bool POOL> ZD uint32_t* extQueue(const Params& P, uint32_t side, uint32_t round) {
    return POOL ? P.q.ext_pool[side][round] : P.q.ext[side];
}
template <bool POOL> ZD QueueCounts* extCount(const Params& P, uint32_t side, uint32_t round) {
    return POOL ? &P.ctl->ext_pool_count[side][round] : &P.ctl->ext_count[side];
}

// the time of the path a queue entry belongs to, for the instantiations that know animated props (ANIM = EXTRA / MESHL)
template <bool POOL, bool ANIM>
ZD uint32_t pathTime(const Params& P, uint32_t vertex) {
    return (ANIM && c_scene.prop_frames) ? P.pool.time[sampleSlot<POOL>(vertex)] : 0u;
}

struct Mediums {  // prop/medium.zig:30-153; an entry is its (prop, part): ior, priority and absorption are the material's
    uint32_t index;
    // push() stops at Num_entries - 1 = 3 entries.  Six words and selects, not arrays: an array that is indexed by a variable stays in
    // scratch, and the record's other members with it
    uint32_t p0, p1, p2, q0, q1, q2;
    float    wavelength;  // probe.wavelength rides in the record's spare word: 0 until a dispersive Glass settles it
    ZD uint32_t prop(uint32_t i) const { return pick(0 == i, p0, pick(1 == i, p1, p2)); }  // (dmath.hpp pick)
    ZD uint32_t part(uint32_t i) const { return pick(0 == i, q0, pick(1 == i, q1, q2)); }
    ZD void     set(uint32_t i, uint32_t fprop, uint32_t fpart) {
        p0 = pick(0 == i, fprop, p0), q0 = pick(0 == i, fpart, q0);
        p1 = pick(1 == i, fprop, p1), q1 = pick(1 == i, fpart, q1);
        p2 = pick(2 == i, fprop, p2), q2 = pick(2 == i, fpart, q2);
    }
    ZD void load(const Params& P, uint32_t vertex) {
        const uint4 a = P.pool.med0[vertex], b = P.pool.med1[vertex];
        index = a.x;
        p0 = a.y, q0 = a.z, p1 = a.w, q1 = b.x, p2 = b.y, q2 = b.z;
        wavelength = __uint_as_float(b.w);
    }
    ZD void store(const Params& P, uint32_t vertex) const {
        P.pool.med0[vertex] = make_uint4(index, p0, q0, p1);
        P.pool.med1[vertex] = make_uint4(q1, p2, q2, __float_as_uint(wavelength));
    }
    ZD const ZygpuMaterial& material(uint32_t i) const {
        return c_scene.materials[c_scene.material_ids[c_scene.props[prop(i)].parts_start + part(i)]];
    }
    ZD float ior(uint32_t i) const {  // material.zig:140-146
        const ZygpuMaterial& m = material(i);
        return (ZYGPU_MAT_GLASS == m.kind || ZYGPU_MAT_SUBSTITUTE == m.kind) ? m.ior : 1.f;
    }
    ZD bool  matches(uint32_t i, uint32_t fprop, uint32_t fpart) const { return prop(i) == fprop && part(i) == fpart; }
    ZD float topIor() const { return index > 0 ? ior(index - 1) : 1.f; }
    ZD float peekIor(uint32_t fprop, uint32_t fpart) const {
        if (index <= 1) return 1.f;
        const uint32_t back = index - 1;
        return matches(back, fprop, fpart) ? ior(back - 1) : ior(back);
    }
    ZD int32_t highestPriority() const {
        int32_t priority = -128;
        for (uint32_t i = 0; i < index; ++i) priority = max(priority, material(i).priority);
        return priority;
    }
    ZD f3 topAbsorption() const {  // topCC: the last entry of the highest priority
        int32_t  priority = -128;
        uint32_t highest = 0;
        for (uint32_t i = 0; i < index; ++i) {
            const int32_t lp = material(i).priority;
            if (lp >= priority) {
                priority = lp;
                highest = i;
            }
        }
        const ZygpuMaterial& m = material(highest);
        return mk3(m.absorption[0], m.absorption[1], m.absorption[2]);
    }
    ZD void push(uint32_t fprop, uint32_t fpart) {
        if (index < 3) {
            set(index, fprop, fpart);
            index += 1;
        }
    }
    ZD void remove(uint32_t fprop, uint32_t fpart) {
        const int32_t back = int32_t(index) - 1;
        for (int32_t i = back; i >= 0; --i) {
            if (matches(uint32_t(i), fprop, fpart)) {
                for (int32_t j = i; j < back; ++j) {
                    set(uint32_t(j), prop(uint32_t(j + 1)), part(uint32_t(j + 1)));
                }
                index -= 1;
                return;
            }
        }
    }
};


ZD uint32_t pixelId(int x, int y) {  // worker.zig:127-128,136,141
    const int fr = c_view.v.filter_radius_int;
    const int r0 = c_view.v.resolution[0] + 2 * fr;
    return uint32_t((y + fr) * r0) + uint32_t(x + fr);
}

ZD uint64_t pcgSequence(uint32_t pixel_id, uint32_t iteration) {  // worker.zig:129-130,143
    const int      fr = c_view.v.filter_radius_int;
    const uint32_t a = uint32_t(c_view.v.resolution[0] + 2 * fr) * uint32_t(c_view.v.resolution[1] + 2 * fr);
    return uint64_t(pixel_id) + uint64_t(iteration) * a;
}

// the iteration whose number selects the pixel's PCG stream: its own (progressive), or the call's first one (one-shot)
ZD uint32_t pcgIteration(uint32_t iteration) { return c_view.one_shot ? c_view.oneshot_base : iteration; }

ZD float splitThreshold(uint32_t total_depth) {  // integrator/helper.zig:33-39
    const float t = c_view.v.split_threshold;
    return mn(total_depth < 4 ? t : kLowThreshold, t);
}
ZD float powerHeuristic(float f_pdf, float g_pdf) {  // helper.zig:64-67
    const float f2 = f_pdf * f_pdf;
    return f2 / __fmaf_rn(g_pdf, g_pdf, f2);
}
ZD float predividedPowerHeuristic(float f_pdf, float g_pdf) {  // helper.zig:70-73
    const float f2 = f_pdf * f_pdf;
    return f_pdf / __fmaf_rn(g_pdf, g_pdf, f2);
}

struct PathHead {  // sampler + packed flags of one path
    Sampler  smp;
    uint32_t depth_surface, vdepth, flags;
    uint32_t nee;  // next-event-estimation state of the current vertex (kNee*)
};

// POOL: `vertex` is a vertex index; the samplers are those of its camera sample, depth and flags are the vertex's own
template <bool POOL = false>
ZD void loadHead(const Params& P, uint32_t vertex, PathHead& h) {
    const uint32_t slot = sampleSlot<POOL>(vertex);
    int      px, py;
    uint32_t iteration;
    slotPixel(POOL ? slot : sampleOf(P, slot), px, py, iteration);
    const uint32_t pixel_id = pixelId(px, py);
    const uint4    a = P.pool.q0[slot];
    const float4   b = P.pool.q1[slot];
    const uint2    g = P.pool.pcg[slot];
    const uint32_t meta = POOL ? ((a.y & 7u) | (P.pool.vmeta.get(vertex) & ~7u)) : a.y;
    h.smp.run_seed = a.x;
    h.smp.buffer[0] = b.x;
    h.smp.buffer[1] = b.y;
    h.smp.buffer[2] = b.z;
    h.smp.buffer[3] = b.w;
    h.smp.buffer[4] = __uint_as_float(a.z);
    h.smp.dimension = meta & 7u;
    const uint64_t sample_index = uint64_t(pixel_id) * uint64_t(c_view.v.num_expected_samples) + uint64_t(iteration);
    h.smp.sample = uint32_t(sample_index);
    h.smp.rng.state = uint64_t(g.x) | (uint64_t(g.y) << 32);
    h.smp.rng.inc = (pcgSequence(pixel_id, pcgIteration(iteration)) << 1) | 1ull;
    h.smp.sobol_kind = ZYGPU_SAMPLER_SOBOL == c_view.v.sampler;
    h.depth_surface = (meta >> 3) & 31u;
    h.vdepth = (meta >> 8) & 31u;
    h.flags = (meta >> 13) & 255u;
    h.nee = a.w;
    h.smp.use_primary = h.depth_surface < 3;  // worker.zig:201-207
}

ZD uint32_t packMeta(const PathHead& h) {
    return (h.smp.dimension & 7u) | ((h.depth_surface & 31u) << 3) | ((h.vdepth & 31u) << 8) | ((h.flags & 255u) << 13);
}

// the samplers of a camera sample (POOL: shared by the vertices of the sample, so stored after every vertex)
// Params::packed_prev: what the next vertex's emission pass needs of this one (vertex.zig:45-65: origin and geometric normal for the
// light's pdf, PathtracerMIS only; the roughness of the event for regularize_roughness) rides in words of the path's first line
// that are dead from the moment a vertex is left (scatter / camera-ray generation: incrementPadding has just made the Sobol buffer
// stale, or the sampler is the PCG stream and never looks at it) until the next shading pass draws again: the five buffer words
// (q1, q0.z) hold origin.xyz, geo_n.xy; geo_n.z sits in the spare lane of the indirect sum, the roughness in the throughput's.
// The first shading pass that stores its samplers overwrites the buffer words, which by then only wf_shade's second pass could
// have wanted, and that pass runs first on the vertices deferred to it.  wf_scatter then writes one line per path instead of two
// (configs[1]: the kernel -8 % as an upper bound).
struct PrevVertex {
    f3    origin, geo_n;
    float alpha;
};
ZD void storeSamplers(const Params& P, uint32_t slot, const PathHead& h, const PrevVertex* prev = nullptr) {
    if (prev) {
        P.pool.q0[slot] = make_uint4(h.smp.run_seed, packMeta(h), __float_as_uint(prev->geo_n.y), h.nee);
        P.pool.q1[slot] = make_float4(prev->origin.x, prev->origin.y, prev->origin.z, prev->geo_n.x);
    } else {
        P.pool.q0[slot] = make_uint4(h.smp.run_seed, packMeta(h), __float_as_uint(h.smp.buffer[4]), h.nee);
        P.pool.q1[slot] = make_float4(h.smp.buffer[0], h.smp.buffer[1], h.smp.buffer[2], h.smp.buffer[3]);
    }
    P.pool.pcg[slot] = make_uint2(uint32_t(h.smp.rng.state), uint32_t(h.smp.rng.state >> 32));
}
// geo_n.z of a packed previous vertex: the spare lane of the indirect sum (IValue.add leaves it alone)
ZD void storePrevTail(const Params& P, uint32_t slot, const PrevVertex& prev) {
    float4 i = P.pool.indirect[slot];
    i.w = prev.geo_n.z;
    P.pool.indirect[slot] = i;
}

template <bool POOL = false>
ZD void storeHead(const Params& P, uint32_t vertex, const PathHead& h, const PrevVertex* prev = nullptr) {
    storeSamplers(P, sampleSlot<POOL>(vertex), h, prev);
    if (POOL) P.pool.vmeta[vertex] = packMeta(h);
}

// The path sampler as the traversal's source of opacity-mask draws (prop.zig:191-195). Each draw loads the sampler from the
// pool and stores it back: held in registers across the traversal loop instead, it cost the EXTRA instantiations 10-16 % on
// scenes without any mask (cornell_glass 129 -> 109, outdoor 249 -> 225 Msamples/s); masked hits are rare, the walk is not.
// Only the EXTRA instantiations carry it; NoHeadDraw is its stand-in elsewhere.
template <bool POOL>
struct HeadDraw {
    const Params& P;
    uint32_t      vertex;
    ZD HeadDraw(const Params& P_, uint32_t vertex_) : P(P_), vertex(vertex_) {}
    ZD float operator()() {
        PathHead h;
        loadHead<POOL>(P, vertex, h);
        const float r = h.smp.sample1D();
        storeHead<POOL>(P, vertex, h);
        return r;
    }
    ZD void flush() const {}
};
struct NoHeadDraw {
    ZD NoHeadDraw(const Params&, uint32_t) {}
    ZD float operator()() const { return 0.f; }
    ZD void flush() const {}
};
template <bool POOL, bool EXTRA>
struct MaskDraw {
    using type = HeadDraw<POOL>;
};
template <bool POOL>
struct MaskDraw<POOL, false> {
    using type = NoHeadDraw;
};

// -DZYG_NAN_TRACE builds (development): a non-finite value at a named site sets bit `site` of Control::error
#ifdef ZYG_NAN_TRACE
ZD void nanTrace(const Params& P, float v, uint32_t site) {
    if (!(fabsf(v) <= FLT_MAX)) atomicOr(&P.ctl->error, 1u << site);
}
ZD void nanTrace(const Params& P, f3 v, uint32_t site) { nanTrace(P, v.x + v.y + v.z, site); }
#else
ZD void nanTrace(const Params&, float, uint32_t) {}
ZD void nanTrace(const Params&, f3, uint32_t) {}
#endif

// An optimisation barrier on values (no instruction): what is computed from them afterwards cannot be moved in front of this point.
// In the kernels that know every light kind LLVM's loop-invariant code motion otherwise hoists the set-up of each kind (SphQuad, cone
// frames, chords, portal windows) and the light-independent half of the closure's evaluation out of the light-sample loop and keeps
// all of it in registers across the shadow ray's walk: wf_scatter<.., MESHL, POOL> 806 spilled registers, 189 with one light kind.
#ifndef ZYG_PIN
#define ZYG_PIN 1
#endif
ZD void pin(float& x) {
#if ZYG_PIN
    asm volatile("" : "+v"(x));
#endif
}
ZD void pin(f3& v) {
    pin(v.x);
    pin(v.y);
    pin(v.z);
}
ZD void pin(uint32_t& x) {
#if ZYG_PIN
    asm volatile("" : "+v"(x));
#endif
}

ZD f3 xyz4(float4 v) { return {v.x, v.y, v.z}; }
ZD float4 f4(f3 v, float w = 0.f) { return make_float4(v.x, v.y, v.z, w); }

// Where the three IValue accumulators of a path live.  POOL (and the builds without the interleaved block): 16 B records of their
// own, emission and direct in a sector of the path's second line.  Otherwise (PACKED): in the six words of the path's first line
// that the non-POOL kernels leave free (40..47: emission.xy; 112..127: emission.z, direct.xyz; POOL has its hit record there):
// the end of a path reads all three sums from the line its kernel read anyway, instead of waiting for a cold line once per wave
// (measured: finalize without these loads at all would be worth 2.8 % on configs[1]; the relocation itself 0 ± 1 % there and
// +1.6 % on configs[2]: -DZYG_NO_PACKED_ACC restores the separate records).
#if ZYG_POOL_STRIDE == 256 && ZYG_HIT_STRIDE && !defined(ZYG_ACC_SEPARATE) && !defined(ZYG_NO_PACKED_ACC)
constexpr bool kPackedAccumulators = true;
ZD char* pathBlock(const Params& P, uint32_t slot) {
    return P.pool.q0.base + size_t((slot & P.pool.q0.mask) + P.pool.q0.local) * size_t(P.pool.q0.stride);
}
#else
constexpr bool kPackedAccumulators = false;
ZD char* pathBlock(const Params&, uint32_t) { return nullptr; }
#endif

// where a record lives (the interleaved block's proxy, or a plain array in the builds without it)
template <typename T, int STRIDE>
ZD T* recordAddress(const Rec<T, STRIDE>& r, uint32_t slot) {
    return r[slot].p;
}
template <typename T>
ZD T* recordAddress(T* r, uint32_t slot) {
    return r + slot;
}

template <bool PACKED>
ZD void loadEmissionDirect(const Params& P, uint32_t slot, f3& e, f3& d) {
    if (PACKED && kPackedAccumulators && P.packed_acc) {
        char*        b = pathBlock(P, slot);
        const float2 lo = *reinterpret_cast<const float2*>(b + 40);
        const float4 hi = *reinterpret_cast<const float4*>(b + 112);
        e = mk3(lo.x, lo.y, hi.x);
        d = mk3(hi.y, hi.z, hi.w);
    } else {
        e = xyz4(P.pool.emission[slot]);
        d = xyz4(P.pool.direct[slot]);
    }
}
template <bool PACKED>
ZD void clearAccumulators(const Params& P, uint32_t slot) {
    if (PACKED && kPackedAccumulators && P.packed_acc) {
        char* b = pathBlock(P, slot);
        *reinterpret_cast<float2*>(b + 40) = make_float2(0.f, 0.f);
        *reinterpret_cast<float4*>(b + 112) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        P.pool.emission[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        P.pool.direct[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    P.pool.indirect[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// IValue.add (helper.zig:11-19) as a read-modify-write of the one accumulator it touches.  Adding an exact zero vector
// is skipped: the accumulators start at +0 and only ever receive non-negative terms, so x + 0 == x bit for bit.
template <bool PACKED>
ZD void ivalueAdd(const Params& P, uint32_t slot, f3 value, uint32_t depth, uint32_t direct_cutoff, bool is_emission, bool singular) {
    if (0.f == value.x && 0.f == value.y && 0.f == value.z) return;
    const bool direct = singular || depth < direct_cutoff;
    if (PACKED && kPackedAccumulators && P.packed_acc && (is_emission || direct)) {
        char*   b = pathBlock(P, slot);
        float4* hi = reinterpret_cast<float4*>(b + 112);
        float4  h = *hi;
        if (is_emission) {
            float2*      lo = reinterpret_cast<float2*>(b + 40);
            const float2 l = *lo;
            *lo = make_float2(l.x + value.x, l.y + value.y);
            h.x += value.z;
        } else {
            h.y += value.x;
            h.z += value.y;
            h.w += value.z;
        }
        *hi = h;
        return;
    }
    const auto   target = is_emission ? P.pool.emission[slot] : direct ? P.pool.direct[slot] : P.pool.indirect[slot];
    const float4 old = target;
    target = f4(xyz4(old) + value, old.w);  // (the indirect sum's spare lane may carry a packed previous vertex's geo_n.z)
}

ZD f3 clampColor(f3 color, float max) {  // sensor.zig:615-624
    const float mc = hmax3(color);
    if (mc > max) return S3(max / mc) * color;
    return color;
}

// end of a path: Sensor.addSample's colour part (sensor.zig:177-181) into the staging buffer
ZD void finalize(const Params& P, uint32_t slot, f3 throughput, uint32_t flags, const Pcg& rng) {
    if (P.pixel_pcg) P.pixel_pcg[slot] = make_uint2(uint32_t(rng.state), uint32_t(rng.state >> 32));  // one-shot: the next sample goes on
    f3 e, d;
    loadEmissionDirect<true>(P, slot, e, d);  // finalize: the non-POOL kernels
    e = clampColor(e, c_view.v.clamp_emission);
    d = clampColor(d, c_view.v.clamp_direct);
    const f3    i = clampColor(xyz4(P.pool.indirect[slot]), c_view.v.clamp_indirect);
    const f3    summed = e + d + i;
    const float alpha = (flags & ST_TRANSPARENT) ? mx(1.f - average3(throughput), 0.f) : 1.f;
    P.stage_color[sampleOf(P, slot)] = f4(summed, alpha);
}

// POOL: a vertex ends without children: Pool.consume's bookkeeping for terminated vertices (vertex.zig:240-262; no shadow
// catchers), done when the vertex ends instead of at the end of its level (same order: vertices end in index order).
// The camera sample ends with its last vertex: result.direct[3] = 0 + transparency[3] (pathtracer_mis.zig:168-170)
ZD void terminateVertex(const Params& P, uint32_t slot, f3 throughput, float split_weight, uint32_t flags, uint32_t vertex = 0) {
    float alpha = P.pool.transp.get(slot);
    f3    through = S3(0.f);  // Pool.transparency's colour: only shadow catchers add to it
    if (P.pool.sc_emis) {     // a scene with shadow catchers (vertex.zig:252-266); `vertex`: the ending vertex' records
        through = xyz4(P.pool.transp_rgb[slot]);
        const f3 emission = xyz4(P.pool.sc_emis[vertex]);
        if (flags & ST_CATCHER_IN_CAMERA) {
            const f3 o = xyz4(P.pool.sc_occ[vertex]), u = xyz4(P.pool.sc_unocc[vertex]);
            const f3 ratio = mk3(o.x < u.x ? o.x / u.x : 1.f, o.y < u.y ? o.y / u.y : 1.f, o.z < u.z ? o.z / u.z : 1.f);
            through = through + ratio * emission * S3(split_weight);
            alpha = alpha + mx((1.f - average3(ratio)) * split_weight, 0.f);
        } else if (flags & ST_TRANSPARENT) {
            alpha = alpha + mx((1.f - average3(throughput)) * split_weight, 0.f);
        } else {
            through = through + emission * S3(split_weight);
            alpha = alpha + split_weight;
        }
        P.pool.transp_rgb[slot] = f4(through);
    } else {
        alpha = (flags & ST_TRANSPARENT) ? alpha + mx((1.f - average3(throughput)) * split_weight, 0.f) : alpha + split_weight;
    }
    P.pool.transp[slot] = alpha;
    const uint32_t st = P.pool.pstate.get(slot) - 1u;
    P.pool.pstate[slot] = st;
    if (0 == (st & 255u)) {
        if (P.pixel_pcg) P.pixel_pcg[slot] = P.pool.pcg[slot];  // one-shot: storeSamplers ran just before (shared samplers)
        const f3 e = clampColor(xyz4(P.pool.emission[slot]), c_view.v.clamp_emission);
        // result.direct += vertices.transparency (pathtracer_mis.zig:168-170), then the sensor's clamp
        const f3 d = clampColor(P.pool.sc_emis ? xyz4(P.pool.direct[slot]) + through : xyz4(P.pool.direct[slot]), c_view.v.clamp_direct);
        const f3 i = clampColor(xyz4(P.pool.indirect[slot]), c_view.v.clamp_indirect);
        P.stage_color[slot] = f4(e + d + i, 0.f + alpha);
    }
}

// (the counting instantiations only.)  Lanes that counted nothing stay away from the counters: the POOL kernels are launched
// over every vertex a level could have, and 4 M idle lanes adding zeros to four words made each wf_intersect<STATS, POOL> launch
// take 25 ms
ZD void flushCounters(const Params& P, int k, const Counters& cnt, unsigned long long rays) {
    if (cnt.nodes) atomicAdd(&P.ctl->k[k].nodes, cnt.nodes);
    if (cnt.tris) atomicAdd(&P.ctl->k[k].tris, cnt.tris);
    if (cnt.instances) atomicAdd(&P.ctl->k[k].instances, cnt.instances);
    if (rays) atomicAdd(&P.ctl->k[k].rays, rays);
}

// whether a new extension ray will enter the bounding box of one of the scene's (largest) mesh props: such rays go to
// the back of the extension queue (see queuePush).  Purely a grouping hint: every ray is traced the same way.
ZD bool entersMeshBounds(f3 origin, f3 direction) {
    const DScene& s = c_scene;
    if (0 == s.num_heavy) return false;
    const Ray ray = Ray::init(origin, direction, 0.f, kRayMaxT);
    bool      hit = false;
    for (uint32_t k = 0; k < s.num_heavy; ++k) hit = hit || aabbIntersect(s.heavy_aabbs[2 * k], s.heavy_aabbs[2 * k + 1], ray);
    return hit;
}

// The camera entity's transformation at a sample's time (scene.propTransformationAt, scene.zig:382-385): Space.frameAt picks the
// keyframe interval and the weight (space.zig:127-137), math.Transformation.lerp interpolates position and scale linearly and
// the rotation with quaternion.slerp (transformation.zig:36-44, quaternion.zig:217-245), ComposedTransformation.init rebuilds
// the rotation rows (composed_transformation.zig:17-29, quaternion.zig:72-116).  Keyframes are one tick (1/60 s) apart.
// `since`: the time in ticks since the start of the frame's first keyframe interval (scene.zig:186-187: what Pool::time holds)
ZD Trafo cameraTrafoSince(const ZygpuViewDesc& v, uint64_t since);
ZD Trafo cameraTrafoAt(const ZygpuViewDesc& v, uint32_t frame, float shutter_sample) {
    const uint64_t tick = 705600000ull / 60;  // space.zig:18-19
    // Base.absoluteTime (camera_base.zig:129-136); no shutter distribution, so the shutter time is the sample (121-127)
    const uint64_t fdi = uint64_t(llround(double(shutter_sample) * double(v.frame_duration)));
    const uint64_t start_time = uint64_t(frame) * v.frame_step;
    return cameraTrafoSince(v, fdi + (start_time % tick));
}
ZD Trafo cameraTrafoSince(const ZygpuViewDesc& v, uint64_t since) {
    const uint64_t tick = 705600000ull / 60;
    uint64_t       i = since / tick;
    float          wgt = float(double(since - i * tick) / double(tick));
    // a shutter sample of exactly 1.0 lands on the last keyframe, where the reference reads one keyframe past its array
    if (i + 1 >= v.camera_num_keyframes) {
        i = v.camera_num_keyframes - 2;
        wgt = 1.f;
    }
    const ZygpuKeyframe& ka = v.camera_keyframes[i];
    const ZygpuKeyframe& kb = v.camera_keyframes[i + 1];
    const f3 t3 = S3(wgt);
    Trafo    out;
    out.position = lerp3(mk3(ka.position[0], ka.position[1], ka.position[2]), mk3(kb.position[0], kb.position[1], kb.position[2]), t3) -
                   mk3(v.camera_origin[0], v.camera_origin[1], v.camera_origin[2]);  // space.zig:121-124
    out.scale = lerp3(mk3(ka.scale[0], ka.scale[1], ka.scale[2]), mk3(kb.scale[0], kb.scale[1], kb.scale[2]), t3);
    const float qa[4] = {ka.rotation[0], ka.rotation[1], ka.rotation[2], ka.rotation[3]};
    float       end[4] = {kb.rotation[0], kb.rotation[1], kb.rotation[2], kb.rotation[3]};
    float       cosom = qa[0] * end[0] + qa[1] * end[1] + qa[2] * end[2] + qa[3] * end[3];
    if (cosom < 0.f) {
        cosom = -cosom;
        for (int k = 0; k < 4; ++k) end[k] = -end[k];
    }
    float sclp, sclq;
    if (1.f - cosom > 0.00001f) {
        const float omega = acosF(cosom), sinom = sinF(omega);
        sclp = sinF((1.f - wgt) * omega) / sinom;
        sclq = sinF(wgt * omega) / sinom;
    } else {
        sclp = 1.f - wgt;
        sclq = wgt;
    }
    float q[4];
    for (int k = 0; k < 4; ++k) q[k] = sclp * qa[k] + sclq * end[k];
    const float tq[4] = {q[0] + q[0], q[1] + q[1], q[2] + q[2], q[3] + q[3]};
    const float xy = tq[1] * q[0], xz = tq[2] * q[0], yz = tq[1] * q[2];
    const float w0 = tq[0] * q[3], w1 = tq[1] * q[3], w2 = tq[2] * q[3];
    const float t0 = (q[3] + q[1]) * (q[3] - q[1]), t1 = (q[0] + q[2]) * (q[0] - q[2]);
    const float t2 = (q[3] + q[0]) * (q[3] - q[0]), t3q = (q[1] + q[2]) * (q[1] - q[2]);
    out.r0 = mk3(t0 + t1, xy - w2, xz + w1);
    out.r1 = mk3(xy + w2, t2 + t3q, yz - w0);
    out.r2 = mk3(xz - w1, yz + w0, t2 - t3q);
    return out;
}

// ---------------------------------------------------------------- wf_generate
// Worker.render seeding (worker.zig:141-149) + Sensor.cameraSample (sensor.zig:152-166) +
// Perspective.generateVertex (camera_perspective.zig:124-150) + Vertex.init (vertex.zig:67-85)
// KEEP (non-POOL): the caller shades the new path at once (wf_trace_shade<FIRST>): the sampler (`keep`) and the ray (`keep_o`,
// `keep_d`) stay in registers, and nothing of the path goes to the pool here: shadeVertex<PRE = 1> writes the records of a path
// that goes on, all at once behind the traversal, and none for a path that ends at its first vertex
template <bool POOL, bool AOV = false, bool KEEP = false>
ZD void generatePath(const Params& P, uint32_t slot, PathHead* keep = nullptr, float4* keep_o = nullptr, float4* keep_d = nullptr,
                     uint32_t sample = kNull) {
    if (kNull == sample) sample = slot;  // (the regenerating kernels name it: the slot is only where the path's records live)
    if (AOV && c_view.v.aov_slots) {  // aov.Value.clear (aov_value.zig:62-73): Depth starts at f32 max
#pragma unroll
        for (uint32_t k = 0; k < 6; ++k) {
            const float d = ZYGPU_AOV_DEPTH == k ? FLT_MAX : 0.f;
            P.stage_aov[k * P.aov_plane + slot] = make_float4(d, d, d, d);
        }
    }

    int      x, y;
    uint32_t iteration;
    slotPixel(sample, x, y, iteration);
    const uint32_t pixel_id = pixelId(x, y);

    PathHead h;
    h.smp.sobol_kind = ZYGPU_SAMPLER_SOBOL == c_view.v.sampler;
    h.smp.use_primary = true;
    const uint32_t nexp = c_view.v.num_expected_samples;
    uint64_t       sample_index = uint64_t(pixel_id) * uint64_t(nexp) + uint64_t(iteration);
    uint32_t       seed = uint32_t(sample_index >> 32) + iteration / nexp;
    if (c_view.one_shot && iteration != c_view.oneshot_base) {
        // sample s > 0 of a one-shot pixel: the stream goes on where the previous sample left it (worker.zig:143, 153-165);
        // Sobol: incrementSample (sobol.zig:26-30) counts the sample index up and returns to the pixel's start seed
        const uint2 st = P.pixel_pcg[slot];
        h.smp.rng.state = uint64_t(st.x) | (uint64_t(st.y) << 32);
        h.smp.rng.inc = (pcgSequence(pixel_id, c_view.oneshot_base) << 1) | 1ull;
        const uint64_t first = uint64_t(pixel_id) * uint64_t(nexp) + uint64_t(c_view.oneshot_base);
        sample_index = uint64_t(uint32_t(first) + (iteration - c_view.oneshot_base));
        seed = uint32_t(first >> 32) + c_view.oneshot_base / nexp;
    } else {
        h.smp.rng.start(0, pcgSequence(pixel_id, iteration));
    }
    h.smp.buffer[0] = h.smp.buffer[1] = h.smp.buffer[2] = h.smp.buffer[3] = h.smp.buffer[4] = 0.f;
    h.smp.startPixel(uint32_t(sample_index), seed);

    float s0, s1, s2, s3;
    h.smp.sample4D(s0, s1, s2, s3);
    // sensor.zig:157: the camera sample's time, through the shutter's distribution where the take configured one
    // (Base.sampleShutterTime, camera_base.zig:121-127 -> Distribution1DN(63).sampleContinous, distribution_1d.zig:188-205)
    const ZygpuViewDesc& v = c_view.v;
    float                shutter_sample = h.smp.sample1D();
    if (v.shutter_configured) {
        const float r = shutter_sample;
        uint32_t    i = 1;
        while (i < 63u && !(v.shutter_cdf[i] >= r)) ++i;  // search(cdf, 1, N, r)
        const uint32_t offset = i - 1u;
        const float    c = v.shutter_cdf[offset + 1u], width = c - v.shutter_cdf[offset];
        shutter_sample = 0.f == width ? 0.f : (float(offset) + (c - r) / width) / 63.f;
    }
    h.smp.incrementPadding();

    const float cx = float(x) + s0, cy = float(y) + s1;
    f3 direction = mk3(v.left_top[0], v.left_top[1], v.left_top[2]) + mk3(v.d_x[0], v.d_x[1], v.d_x[2]) * S3(cx) +
                   mk3(v.d_y[0], v.d_y[1], v.d_y[2]) * S3(cy);
    f3 origin;
    if (v.lens_radius > 0.f) {
        float lx, ly;
        if (kNull != v.aperture_shape) {  // Aperture.sample with a shape (aperture.zig:46-53): 2 * sampleContinuous(uv).uv - 1
            float u, w;
            imageLightSample(c_scene.image_lights[v.aperture_shape], s2, s3, u, w);
            lx = 2.f * u - 1.f;
            ly = 2.f * w - 1.f;
        } else {
            diskConcentric(s2, s3, lx, ly);
        }
        origin = mk3(lx * v.lens_radius, ly * v.lens_radius, 0.f);
        const f3 focus = S3(v.focus_distance / direction.z) * direction;
        direction = focus - origin;
    } else {
        origin = mk3(v.eye_offset[0], v.eye_offset[1], v.eye_offset[2]);
    }
    Trafo ct;
    if (v.camera_num_keyframes > 1) {  // an animated camera: its transformation at the sample's time
        ct = cameraTrafoAt(v, c_view.frame, shutter_sample);
    } else {
        ct.r0 = mk3(v.camera_trafo.r[0][0], v.camera_trafo.r[0][1], v.camera_trafo.r[0][2]);
        ct.r1 = mk3(v.camera_trafo.r[1][0], v.camera_trafo.r[1][1], v.camera_trafo.r[1][2]);
        ct.r2 = mk3(v.camera_trafo.r[2][0], v.camera_trafo.r[2][1], v.camera_trafo.r[2][2]);
        ct.scale = mk3(v.camera_trafo.r[0][3], v.camera_trafo.r[1][3], v.camera_trafo.r[2][3]);
        ct.position = mk3(v.camera_trafo.position[0], v.camera_trafo.position[1], v.camera_trafo.position[2]);
    }
    const f3 origin_w = ct.objectToWorldPoint(origin);
    const f3 direction_w = ct.objectToWorldVector(normalize3(direction));

    h.depth_surface = 0;
    h.vdepth = 0;
    h.flags = ST_PRIMARY | ST_TRANSPARENT | ST_SINGULAR;
    h.nee = 0;
    if (KEEP) {
        *keep = h;
        *keep_o = f4(origin_w, kRayMaxT);
        *keep_d = f4(direction_w, 0.f);
    } else {
        const bool       packed = !POOL && 0 != P.packed_prev;
        const PrevVertex camera{origin_w, S3(0.f), 0.f};  // Vertex.init (vertex.zig:67-85): origin = the ray's, geo_n = 0
        storeHead<POOL>(P, slot, h, packed ? &camera : nullptr);  // POOL: vertex 0 of half 0 has the slot's own index
        P.pool.thr[slot] = make_float4(1.f, 1.f, 1.f, POOL ? 1.f : 0.f);              // POOL: w = vertex.split_weight
        P.pool.ray_o[slot] = f4(origin_w, kRayMaxT);
        P.pool.ray_d[slot] = f4(direction_w, 0.f);
        if (!packed) P.pool.vorigin[slot] = f4(origin_w, POOL ? __uint_as_float(1u) : 0.f);  //       w = vertex.path_count
        if (POOL) {
            P.pool.med0[slot] = make_uint4(0u, 0u, 0u, 0u);
            P.pool.med1[slot] = make_uint4(0u, 0u, 0u, 0u);  // .w: probe.wavelength = 0
            P.pool.pstate[slot] = 1u;
            P.pool.transp[slot] = 0.f;
            if (P.pool.sc_emis) {  // Vertex.init (vertex.zig:78-80), Pool.start (vertex.zig:229)
                P.pool.sc_emis[slot] = P.pool.sc_occ[slot] = P.pool.sc_unocc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                P.pool.transp_rgb[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (!packed) P.pool.vgeo_n[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        clearAccumulators<!POOL>(P, slot);  // (the indirect sum's spare lane = 0 = the packed geo_n.z)
    }
    P.stage_uv[sample] = make_float2(s0, s1);
    // AOV = the MESHL instantiations: Probe.time for the animated props (camera_base.zig:129-136), and for the camera's transformation
    // at a vertex where a procedural texture asks for the pixel's footprint (context.zig:140)
    if (AOV && (c_scene.prop_frames || (c_scene.procedurals && v.camera_num_keyframes > 1))) {
        const uint64_t tick = 705600000ull / 60, start_time = uint64_t(c_view.frame) * v.frame_step;
        const uint64_t fdi = uint64_t(llround(double(shutter_sample) * double(v.frame_duration)));
        P.pool.time[slot] = uint32_t(fdi + (start_time % tick));
    }
}

template <bool POOL>
__global__ void __launch_bounds__(256) wf_generate(Params P, uint32_t num_paths) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const bool     valid = slot < num_paths;
    if (0 == slot) P.ctl->bound[1] = num_paths;  // tick 0's wf_intersect reads the previous tick's bound
    if (valid) generatePath<POOL, true>(P, slot);
    const bool back = valid && entersMeshBounds(xyz4(P.pool.ray_o[slot]), xyz4(P.pool.ray_d[slot]));
    queuePush(extQueue<POOL>(P, P.parity, 0), extCount<POOL>(P, P.parity, 0), P.queue_capacity, valid, slot, back);
}

// PropBvh.emission over the unoccluding TLAS (prop_tree.zig:302-356, prop.zig:239-264, rectangle.zig:188-196), split in
// two: wf_intersect walks the tree right after the closest hit (ray.max_t already shrunk to it) and records the lights
// the ray crosses, in traversal order; wf_shade evaluates them (sampler draws + MIS weights) in that order.  The list holds
// kMaxUnoccludingCandidates crossings; a ray that crosses more (a row of lamps seen end-on: "Light" entities are unoccluding by
// default, scene_loader.zig:249) has its emission evaluated by a second walk in the shading pass, in the same order.
constexpr uint32_t kMaxUnoccludingCandidates = 4;

// whether connectLight looks at emission at all for this vertex, and at the unoccluding lights in particular
// (pathtracer.zig:110-143, pathtracer_dl.zig:173-207, pathtracer_mis.zig:280-341)
ZD bool connectsLights(uint32_t flags) { return !(!c_view.v.caustics_path && (flags & ST_SPECULAR) && !(flags & ST_PRIMARY)); }
ZD bool walksUnoccluding(uint32_t flags) {
    return connectsLights(flags) && (ZYGPU_INTEGRATOR_PTDL != c_view.v.integrator || (flags & ST_SINGULAR));
}

// The walk itself: every unoccluding emitter the ray crosses in front of max_t, in traversal order, handed to
// record(prop, primitive, t, u, v).
template <bool STATS, bool EXTRA, bool UTLAS = false, typename Record>
ZD void walkUnoccluding(const DScene& s, uint32_t depth_surface, const Ray& ray_in, Counters& cnt, Record&& record) {
    Ray ray = ray_in;
    // the unoccluding tree is one leaf of Rectangles (dscene.hpp TlasEntry): its order is the leaf's order; the complete kernels of the
    // relaxed units take the table too where the scene's tables stand in for its trees (DScene::table_complete)
    if (UTLAS || (ZYG_FAST && !STATS && EXTRA && s.table_complete)) {
        for (uint32_t k = 0; k < s.num_unocc_table; ++k) {
            const TlasEntry& e = s.unocc_table[k];
            if (!propVisibleAtDepth(e.flags, depth_surface)) continue;
            if (!entryBoundsHit<0 != ZYG_FAST>(e, ray)) continue;
            float t, u, v;
            if ((ZYG_FAST && ZYG_RECT_UNIT) ? unitRectangleIntersect(ray, e, t, u, v) : rectangleIntersect(ray, entryTrafo(e), t, u, v)) record(e.prop, 0, t, u, v);
        }
        return;
    }
    bvhWalk<STATS>(s.unocc_nodes, s.num_unocc_nodes, ray, cnt, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            const uint32_t  p = s.unocc_indices[i];
            const ZygpuProp prop = s.props[p];
            if (!propVisibleAtDepth(prop.flags, depth_surface)) continue;
            if (STATS) cnt.instances += 1;
            if (!aabbIntersect(s.aabbs[2 * p], s.aabbs[2 * p + 1], ray)) continue;
            const Trafo trafo = propTrafoAt<EXTRA, true>(s, p, ray.time);
            if (ZYGPU_SHAPE_RECTANGLE == prop.shape) {
                float t, u, v;
                if (rectangleIntersect(ray, trafo, t, u, v)) record(p, 0, t, u, v);
            } else if (EXTRA && ZYGPU_SHAPE_SPHERE == prop.shape) {  // Sphere.emission (sphere.zig:269-277)
                float t;
                if (sphereIntersect(ray, trafo, t)) record(p, 0, t, 0.f, 0.f);
            } else if (EXTRA && ZYGPU_SHAPE_DISK == prop.shape) {  // Disk.emission (disk.zig:171-179)
                float t, u, v;
                if (diskIntersect(ray, trafo, t, u, v)) record(p, 0, t, u, v);
            } else if (EXTRA && ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
                // TriangleTree.emission (triangle_tree.zig:405-483): every triangle the ray crosses, no max_t shrinking
                const DMesh& mesh = s.meshes[prop.mesh];
                Ray          local = worldToObjectRay(trafo, ray);
                bvhWalk<STATS>(mesh.nodes, 1u, local, cnt, [&](uint32_t tfirst, uint32_t tcount) {
                    for (uint32_t k = tfirst, ke = tfirst + tcount; k < ke; ++k) {
                        if (STATS) cnt.tris += 1;
                        const f3 a = xyz(mesh.tri_verts[3 * k]), b = xyz(mesh.tri_verts[3 * k + 1]), c = xyz(mesh.tri_verts[3 * k + 2]);
                        float    t, u, v;
                        if (triangleIntersect(local, a, b, c, t, u, v)) record(p, k, t, u, v);
                    }
                    return false;
                });
            }
        }
        return false;
    });
}

// The traversal kernels' half: the first kMaxUnoccludingCandidates crossings go into the path's candidate list; the count
// returned is the number of crossings, which may be larger: the shading pass then walks the tree again itself (unoccludingEmission).
template <bool STATS, bool EXTRA, bool UTLAS = false>
ZD uint32_t unoccludingCandidates(const Params& P, const DScene& s, uint32_t slot, size_t plane, uint32_t depth_surface, const Ray& ray_in,
                                  Counters& cnt) {
    uint32_t n = 0;
    walkUnoccluding<STATS, EXTRA, UTLAS>(s, depth_surface, ray_in, cnt, [&](uint32_t prop, uint32_t primitive, float t, float u, float v) {
        if (n < kMaxUnoccludingCandidates) {
            const size_t at = size_t(n) * plane + slot;
            P.pool.ucand[at] = make_uint4(prop, __float_as_uint(t), __float_as_uint(u), __float_as_uint(v));
            P.pool.ucand_prim[at] = primitive;
        }
        n += 1;
    });
    return n;
}

// ---------------------------------------------------------------- wf_intersect
// EXTRA (= the scenes that run the MESHL instantiations): Sphere / Disk / Cube code (dscene.hpp), mesh lights as unoccluding
// candidates
template <bool STATS, bool POOL, bool EXTRA, bool WIDEC = false>
__global__ void __launch_bounds__(256) wf_intersect(Params P) {
    const uint32_t cur = P.parity;
    // the previous tick's queue was no longer (POOL: a level can have more vertices than the one before)
    if (!POOL && blockIdx.x * blockDim.x >= P.ctl->bound[cur ^ 1u]) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at;
    const uint32_t n = queueOpen(extCount<POOL>(P, cur, P.round), P.queue_capacity, i, at);
    if (0 == i) P.ctl->bound[cur] = n;
    if (0 == blockIdx.x && threadIdx.x < 64) {  // this tick's wf_shade / wf_nee / wf_extend fill these
        if (!POOL) queueReset(&P.ctl->ext_count[cur ^ 1u]);
        if (POOL && 0 == P.round)
            for (uint32_t k = 0; k < kPoolVertices; ++k) queueReset(&P.ctl->ext_pool_count[cur ^ 1u][k]);
        aliveReset(P.ctl->alive_count[cur]);
        queueReset(&P.ctl->shadow_count[0]);
        queueReset(&P.ctl->shadow_count[1]);
        queueReset(&P.ctl->heavy_count[cur]);
    }
    Counters cnt{0, 0, 0};
    if (i < n) {
        const uint32_t slot = extQueue<POOL>(P, cur, P.round)[at];  // POOL: a vertex index
        const float4   o = P.pool.ray_o[slot], d = P.pool.ray_d[slot];
        const uint32_t meta = POOL ? P.pool.vmeta.get(slot) : P.pool.q0.get(slot).y;
        const uint32_t depth = (meta >> 3) & 31u, flags = (meta >> 13) & 63u;
        Ray            ray = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
        ray.time = pathTime<POOL, EXTRA>(P, slot);
        if (POOL) {
            // inside a dielectric the ray ends where it leaves the medium prop's bounding box
            // (VolumeIntegrator.integrate, volume_integrator.zig:98-103); wf_shade sees the shortened ray
            Mediums md;
            md.load(P, slot);
            if (md.index > 0) {
                const uint32_t mp = md.prop(md.index - 1);
                const float    limit = aabbIntersectP(c_scene.aabbs[2 * mp], c_scene.aabbs[2 * mp + 1], ray);
                ray.max_t = mn(offsetF(limit), o.w);
                P.pool.ray_o[slot] = f4(ray.origin, ray.max_t);
            }
        }
        Hit            hit;
        typename MaskDraw<POOL, EXTRA>::type draw(P, slot);
        sceneIntersect<STATS, EXTRA, WIDEC>(c_scene, ray, depth, hit, cnt, draw);
        draw.flush();
        if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
        P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
        P.pool.hit_prop[slot] = hit.prop;
        // after Scene.intersect the probe's max_t is the hit distance (prop_tree.zig:76-79)
        P.pool.ucount[slot] = walksUnoccluding(flags) ? unoccludingCandidates<STATS, EXTRA>(P, c_scene, slot, vertexPlane<POOL>(), depth, ray, cnt) : 0u;
    }
    if (STATS) flushCounters(P, 0, cnt, (0 == blockIdx.x && 0 == threadIdx.x) ? n : 0);
}

// ---------------------------------------------------------------- wf_shade

ZD float shapeArea(const DScene& s, uint32_t prop, uint32_t part, const Trafo& trafo) {  // shape.zig:132-152
    const ZygpuProp p = s.props[prop];
    if (ZYGPU_SHAPE_RECTANGLE == p.shape) return trafo.scale.x * trafo.scale.y;
    if (ZYGPU_SHAPE_DISTANT == p.shape) return distantSolidAngle(trafo.scale.x);  // "the solid angle, not the area" (shape.zig:140-142)
    if (ZYGPU_SHAPE_SPHERE == p.shape) return (4.f * kPi) * ((0.5f * trafo.scale.x) * (0.5f * trafo.scale.x));  // shape.zig:148
    if (ZYGPU_SHAPE_DOME == p.shape) return 4.f * kPi;
    if (ZYGPU_SHAPE_CANOPY == p.shape) return 2.f * kPi;
    if (ZYGPU_SHAPE_DISK == p.shape) return kPi * ((0.5f * trafo.scale.x) * (0.5f * trafo.scale.x));
    if (ZYGPU_SHAPE_CUBE == p.shape) return 2.f * (trafo.scale.x * trafo.scale.y + trafo.scale.x * trafo.scale.z + trafo.scale.y * trafo.scale.z);
    return s.meshes[p.mesh].part_areas[part] * (trafo.scale.x * trafo.scale.y);
}

// SkyMaterial.evaluateRadiance without a map (sky_material.zig:150-168): the sun's radiance over its disc,
// sun_radiance.eval(sunV(rs.trafo.rotation, wi)): InterpolatedFunction1D.eval over [0, 1] (interpolated_function.zig:35-46)
ZD f3 sunRadiance(const DScene& s, const Trafo& trafo, f3 wi) {
    const float radius = 0.0046548266f;  // Sky.Radius = tan(0.2667 degrees) as f32 (sky.zig:46; ArPragueSkyModelGround.h:88)
    const f3    k = wi - trafo.r2;
    const float c = -dot3(trafo.r1, k) / radius;
    const float x = mx((c + 1.f) * 0.5f, 0.f);
    const float inverse_interval = 1.f / (1.f / float(s.num_sun_radiance - 1u));
    const float cx = mn(x, 1.f);
    const float o = cx * inverse_interval;
    const uint32_t offset = uint32_t(o);
    const float    t = o - float(offset);
    const f3 a = xyz4(s.sun_radiance[offset]), b = xyz4(s.sun_radiance[min(offset + 1u, s.num_sun_radiance - 1u)]);
    return lerp3(a, b, S3(t));
}

// Material.evaluateRadiance for Light / Substitute (material.zig:196-212, light_material.zig:131-140)
// TEX (= the MESHL instantiations): emission maps, addressed by (u, v) with the stochastic filter's r
template <bool TEX = false>
ZD f3 materialRadiance(const DScene& s, const ZygpuMaterial& m, f3 wi, const Trafo& trafo, uint32_t prop, uint32_t part,
                       bool in_camera, float u = 0.f, float v = 0.f, float r = 0.f, f3 origin = f3{0.f, 0.f, 0.f}) {
    if (TEX && ZYGPU_MAT_SKY == m.kind) return sunRadiance(s, trafo, wi);  // the Sky's sun (the sky itself is a Light with a map)
    if (ZYGPU_MAT_LIGHT != m.kind && ZYGPU_MAT_SUBSTITUTE != m.kind) return S3(0.f);
    const float area = m.emission_normalize ? shapeArea(s, prop, part, trafo) : 1.f;
    return emittanceRadiance<TEX>(s, m, wi, trafo, area, in_camera, u, v, r, origin);
}

// The light kinds of the MESHL instantiations' light loop (one sampling routine each).  KINDS: the kinds an instantiation compiles in
// (bit 1 << kind); wf_light exists for the sets scenes mostly have (Rectangle lights only; a Sky entity's sun and sky map; all), and
// zygpu_render launches the smallest one that holds the scene's kinds (lightKindOf below, the same decision on the host).
enum LightKind : uint32_t { LK_NONE, LK_PORTAL, LK_IMAGE_FLAT, LK_IMAGE_SPHERE, LK_IMAGE_CANOPY, LK_MESH, LK_INFINITE, LK_SPHERE, LK_DISK, LK_RECTANGLE };
constexpr uint32_t kAllLightKinds = 0xFFFFFFFFu;
constexpr uint32_t kRectangleLightKinds = 1u << LK_RECTANGLE, kSkyLightKinds = (1u << LK_INFINITE) | (1u << LK_IMAGE_CANOPY);
inline uint32_t lightKindOf(uint32_t light_class, uint32_t sampler, uint32_t shape) {  // (host: zygpu_upload_scene)
    if (ZYGPU_LIGHT_PORTAL_IMAGE == light_class) return LK_PORTAL;
    if (ZYGPU_LIGHT_PROP_IMAGE == light_class)
        return (ZYGPU_SHAPE_DOME == shape || ZYGPU_SHAPE_RECTANGLE == shape || ZYGPU_SHAPE_DISK == shape) ? LK_IMAGE_FLAT
               : ZYGPU_SHAPE_SPHERE == shape                                                           ? LK_IMAGE_SPHERE
                                                                                                       : LK_IMAGE_CANOPY;
    if (0xFFFFFFFFu != sampler) return LK_MESH;
    if (ZYGPU_SHAPE_DISTANT == shape || ZYGPU_SHAPE_DOME == shape || ZYGPU_SHAPE_CANOPY == shape) return LK_INFINITE;
    return ZYGPU_SHAPE_SPHERE == shape ? LK_SPHERE : ZYGPU_SHAPE_DISK == shape ? LK_DISK : LK_RECTANGLE;
}
struct VertexIn {  // what connectLight needs of the previous vertex (vertex.zig:45-65)
    f3       vorigin, vgeo_n;
    float    bxdf_pdf, light_split_threshold;
    uint32_t flags, depth_surface;
};

// Scene.lightPdf (scene.zig:624-634); MESHL: the scene has TriangleMesh lights
template <bool MESHL, uint32_t KINDS = kAllLightKinds>
ZD float lightPdf(const DScene& s, const VertexIn& v, f3 ray_direction, const Fragment& frag) {
    constexpr auto has = [](uint32_t kind) { return 0 != (KINDS & (1u << kind)); };  // the light kinds compiled into this instantiation
    const uint32_t light_id = s.light_ids[s.props[frag.prop].parts_start + frag.part];
    if ((v.flags & ST_SINGULAR) || kNull == light_id) return 1.f;
    const LightTree  lt{s, s.light_tree};
    const bool       translucent = 0 != (v.flags & ST_TRANSLUCENT);
    const float      select_pdf = lt.pdf(v.vorigin, v.vgeo_n, translucent, v.light_split_threshold, light_id);
    const ZygpuLight light = s.lights[light_id];
    float            sample_pdf;
    if (MESHL && has(LK_PORTAL) && ZYGPU_LIGHT_PORTAL_IMAGE == light.light_class) {  // light.zig:376-379 -> shape.zig:494-506 -> rectangle.zig:597-611
        sample_pdf = rectanglePortalPdf(s.portal_lights[light.sampler], ray_direction, v.vorigin, frag.trafo,
                                        float(numSamples(light.num_samples, v.light_split_threshold)));
    } else if (MESHL && !has(LK_IMAGE_FLAT) && !has(LK_IMAGE_SPHERE) && has(LK_IMAGE_CANOPY) && ZYGPU_LIGHT_PROP_IMAGE == light.light_class) {
        sample_pdf = imageLightPdf(s.image_lights[light.sampler], frag.uv0, frag.uv1) / (2.f * kPi);  // (a Canopy's map is the only image light here)
    } else if (MESHL && (has(LK_IMAGE_FLAT) || has(LK_IMAGE_SPHERE) || has(LK_IMAGE_CANOPY)) && ZYGPU_LIGHT_PROP_IMAGE == light.light_class) {  // light.zig:371-374 -> shape.zig:509-526
        const DImageLight& il = s.image_lights[light.sampler];
        const uint32_t image_shape = s.props[light.prop].shape;
        sample_pdf = ZYGPU_SHAPE_DOME == image_shape ? domeMaterialPdf(il, frag.uv0, frag.uv1, numSamples(light.num_samples, v.light_split_threshold))
                     : ZYGPU_SHAPE_RECTANGLE == image_shape  // rectangle.zig:577-595
                         ? rectangleMaterialPdf(il, ray_direction, v.vorigin, frag.trafo, frag.p, frag.uv0, frag.uv1,
                                                numSamples(light.num_samples, v.light_split_threshold))
                     : ZYGPU_SHAPE_DISK == image_shape  // disk.zig:535-553
                         ? diskMaterialPdf(il, ray_direction, v.vorigin, frag.trafo, frag.p, frag.uv0, frag.uv1,
                                           numSamples(light.num_samples, v.light_split_threshold))
                     : ZYGPU_SHAPE_SPHERE == image_shape  // sphere.zig:489-502
                         ? sphereMaterialPdf(il, ray_direction, v.vorigin, frag.trafo, frag.p, frag.geo_n, frag.uv0, frag.uv1)
                         : imageLightPdf(il, frag.uv0, frag.uv1) / (2.f * kPi);
    } else if (MESHL && has(LK_MESH) && kNull != light.sampler) {  // light.zig:149-157 -> shape.zig:469-492 -> triangle_mesh.zig:647-691
        sample_pdf = meshLightPdf(s, s.mesh_lights[light.sampler], ray_direction, v.vorigin, v.vgeo_n, frag.trafo, frag.p, frag.geo_n,
                                  frag.primitive, translucent, v.light_split_threshold);
    } else if (MESHL && has(LK_INFINITE) && ZYGPU_SHAPE_DISTANT == s.props[light.prop].shape) {  // -> distant.zig:139-141
        sample_pdf = 1.f / distantSolidAngle(frag.trafo.scale.x);
    } else if (MESHL && has(LK_SPHERE) && ZYGPU_SHAPE_SPHERE == s.props[light.prop].shape) {  // -> sphere.zig:469-485
        sample_pdf = spherePdf(v.vorigin, frag.trafo, numSamples(light.num_samples, v.light_split_threshold));
    } else if (MESHL && has(LK_DISK) && ZYGPU_SHAPE_DISK == s.props[light.prop].shape) {  // -> disk.zig:492-533
        sample_pdf = diskPdf(ray_direction, v.vorigin, frag.trafo, frag.p, numSamples(light.num_samples, v.light_split_threshold));
    } else if (MESHL && has(LK_INFINITE) && ZYGPU_SHAPE_DOME == s.props[light.prop].shape) {  // -> dome.zig:214-220
        sample_pdf = translucent ? 1.f / (4.f * kPi) : 1.f / (2.f * kPi);
    } else if (MESHL && has(LK_INFINITE) && ZYGPU_SHAPE_CANOPY == s.props[light.prop].shape) {  // shape.zig:483
        sample_pdf = 1.f / (2.f * kPi);
    } else if (has(LK_RECTANGLE)) {  // -> rectangle.zig:554-575, with the light's sample count under the previous vertex's split threshold
        sample_pdf = rectanglePdf(v.vorigin, frag.trafo, numSamples(light.num_samples, v.light_split_threshold));
    } else {
        sample_pdf = 0.f;  // (no light of this instantiation's kinds: zygpu_render launches the instantiation that holds the scene's)
    }
    return powerHeuristic(v.bxdf_pdf, sample_pdf * select_pdf);
}

// Vertex.evaluateRadiance (vertex.zig:183-212)
template <bool MESHL, uint32_t KINDS = kAllLightKinds>
ZD f3 evaluateRadiance(const DScene& s, const VertexIn& v, f3 wo, const Fragment& frag, Sampler& smp) {
    const ZygpuMaterial& m = s.materials[s.material_ids[s.props[frag.prop].parts_start + frag.part]];
    if (!(m.flags & ZYGPU_MATF_EMISSIVE) || (!(m.flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo))) return S3(0.f);
    // rs.uvw = scene.lightPortalUvw(vertex, frag) (scene.zig:636-648): through a Portal the map is read where the ray points
    float u = frag.uv0, vv = frag.uv1;
    if (MESHL && 0 != (KINDS & (1u << LK_PORTAL)) && s.portal_lights) {
        const uint32_t light_id = s.light_ids[s.props[frag.prop].parts_start + frag.part];
        if (kNull != light_id && ZYGPU_LIGHT_PORTAL_IMAGE == s.lights[light_id].light_class) {
            const DPortalLight& pl = s.portal_lights[s.lights[light_id].sampler];
            portalUvw(loadTrafo(s, s.lights[pl.light_link].prop), -wo, u, vv);  // (the host refuses an animated link)
        }
    }
    const float stochastic_r = smp.sample1D();  // rs.stochastic_r
    const f3    energy = materialRadiance<MESHL>(s, m, wo, frag.trafo, frag.prop, frag.part, 0 == v.depth_surface, u, vv, stochastic_r, v.vorigin);
    const float weight = lightPdf<MESHL, KINDS>(s, v, -wo, frag);
    return S3(weight) * energy;
}

// the infinite props a ray reaches when it leaves the scene: Context.propIntersect + the shape's intersect / fragment
// (prop.zig:163-197; distant.zig:22-74, dome.zig:26-56, canopy.zig:27-58), then Vertex.evaluateRadiance;
// `only_non_lights`: PathtracerDL's filter
// `catcher` (PathtracerMIS on or behind a shadow catcher, pathtracer_mis.zig:303-337; `ray` then reaches to RayMaxT whatever was
// hit): what a shadow-catcher light sends goes to catcher[1] (unoccluded) and, if the ray really left the scene (`escaped`), to
// catcher[0] (occluded) as well, instead of into the returned emission
template <bool STATS, uint32_t KINDS = kAllLightKinds>
ZD f3 infiniteEmission(const DScene& s, const VertexIn& v, const Ray& ray, Sampler& smp, bool only_non_lights, Counters& cnt,
                       f3* catcher = nullptr, bool escaped = true) {
    f3       energy = S3(0.f);
    const f3 wo = -ray.direction;
    for (uint32_t i = 0; i < s.num_infinite_props; ++i) {
        const uint32_t  p = s.infinite_props[i];
        const ZygpuProp prop = s.props[p];
        if (!propVisibleAtDepth(prop.flags, v.depth_surface)) continue;
        if (STATS) cnt.instances += 1;
        if (!aabbIntersect(s.aabbs[2 * p], s.aabbs[2 * p + 1], ray)) continue;
        Fragment frag;
        frag.trafo = propTrafoAt<true, true>(s, p, ray.time);  // an infinite light may be animated like any prop (prop.zig:239-264)
        float u = 0.f, vv = 0.f;
        if (ZYGPU_SHAPE_DISTANT == prop.shape) {
            if (!distantIntersect(ray, frag.trafo, u, vv)) continue;
            frag.geo_n = frag.n = frag.trafo.r2;
            frag.uv0 = (u + 1.f) * 0.5f;
            frag.uv1 = (vv + 1.f) * 0.5f;
        } else {  // the whole sphere / the hemisphere around the prop's z axis
            if (ray.max_t < kRayMaxT) continue;
            if (ZYGPU_SHAPE_CANOPY == prop.shape && dot3(ray.direction, frag.trafo.r2) < kCanopyEps) continue;
            frag.geo_n = frag.n = -ray.direction;
            if (ZYGPU_SHAPE_DOME == prop.shape) domeUv(ray.direction, frag.trafo, frag.uv0, frag.uv1);
            else canopyUv(ray.direction, frag.trafo, frag.uv0, frag.uv1);
        }
        frag.prop = p;
        frag.t_hit = kRayMaxT;
        frag.u = u;
        frag.v = vv;
        frag.primitive = 0;
        frag.p = S3(kRayMaxT) * ray.direction;
        frag.t = frag.trafo.r0;
        frag.b = frag.trafo.r1;
        frag.part = 0;
        if (only_non_lights && !((v.flags & ST_SINGULAR) || kNull == s.light_ids[prop.parts_start])) continue;
        const f3 local_energy = evaluateRadiance<true, KINDS>(s, v, wo, frag, smp);
        if (catcher && (prop.flags & ZYGPU_PROP_SHADOW_CATCHER_LIGHT)) {
            if (escaped) catcher[0] = catcher[0] + local_energy;  // "The light was not occluded"
            catcher[1] = catcher[1] + local_energy;               // otherwise: "occluded, but we forced a hit"
        } else {
            energy = energy + local_energy;
        }
    }
    return energy;
}

// the evaluation half of PropBvh.emission: the candidates wf_intersect recorded, in its traversal order
template <bool MESHL, uint32_t KINDS = kAllLightKinds>
ZD f3 unoccludingEmission(const Params& P, const DScene& s, uint32_t slot, size_t plane, const VertexIn& v, const Ray& ray, Sampler& smp) {
    f3             energy = S3(0.f);
    const uint32_t n = P.pool.ucount[slot];
    const f3       wo = -ray.direction;
    auto           evaluate = [&](uint32_t prop_id, uint32_t prim, float t, float fu, float fv) {
        const ZygpuProp prop = s.props[prop_id];
        Fragment        frag;
        frag.trafo = propTrafoAt<MESHL, true>(s, prop_id, ray.time);
        frag.prop = prop_id;
        frag.t_hit = t;
        frag.u = fu;
        frag.v = fv;
        if (MESHL && ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
            // the partial fragment TriangleTree.emission fills (triangle_tree.zig:428-447): p, geometric normal, part
            const DMesh& mesh = s.meshes[prop.mesh];
            const f3     a = xyz(mesh.tri_verts[3 * prim]), b = xyz(mesh.tri_verts[3 * prim + 1]), cc = xyz(mesh.tri_verts[3 * prim + 2]);
            frag.primitive = prim;
            frag.part = mesh.triangle_parts[prim];
            frag.p = frag.trafo.objectToWorldPoint(interpolate3(a, b, cc, frag.u, frag.v));
            frag.geo_n = frag.trafo.objectToWorldNormal(normalize3(cross3(b - a, cc - a)));
        } else if (MESHL && ZYGPU_SHAPE_SPHERE == prop.shape) {
            frag.primitive = 0;
            sphereFragment(ray, frag);
        } else if (MESHL && ZYGPU_SHAPE_DISK == prop.shape) {
            frag.primitive = 0;
            diskFragment(ray, frag);
        } else {
            frag.primitive = 0;
            rectangleFragment(ray, frag);
        }
        energy = energy + evaluateRadiance<MESHL, KINDS>(s, v, wo, frag, smp);
    };
    if (n > kMaxUnoccludingCandidates) {
        // more crossings than the list holds: walk the tree again, as PropBvh.emission itself does (prop_tree.zig:302-356), and
        // evaluate every crossing where it is found; `ray` ends at the hit like the traversal kernel's.  Nothing is counted twice.
        Counters none{0, 0, 0};
        walkUnoccluding<false, MESHL>(s, v.depth_surface, ray, none, evaluate);
        return energy;
    }
    for (uint32_t k = 0; k < n; ++k) {
        const size_t at = size_t(k) * plane + slot;
        const uint4  c = P.pool.ucand[at];
        evaluate(c.x, MESHL ? P.pool.ucand_prim[at] : 0u, __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w));
    }
    return energy;
}

// One vertex of wf_shade: connectLight, termination, Russian roulette, Vertex.sample's record (pathtracer_mis.zig:45-92,
// 280-341, vertex.zig:137-212).  The ray, the hit and the candidate count come from the caller: wf_shade reads them from the
// pool, wf_trace_shade still has them in registers.  `defer` (first pass only): the vertex has emission to evaluate and
// goes to the second pass untouched.
// PRE: the caller holds the path's head (`pre`) and throughput (`pre_thr`).  1: a path generated in this kernel, nothing of it is
// in the pool yet; 2: read from the pool before the traversal, so that the loads do not wait behind it
template <bool STATS, bool MESHL, bool HEAVY, bool POOL, int PRE = 0, uint32_t KINDS = kAllLightKinds>
ZD void shadeVertex(const Params& P, uint32_t slot, float4 ro, float4 rd, const Hit& hit, uint32_t ucount, bool& alive, bool& defer,
                    Counters& cnt, SectionTimer& tm, const PathHead* pre = nullptr, float4 pre_thr = make_float4(1.f, 1.f, 1.f, 0.f),
                    float* sr_out = nullptr, uint32_t* material_out = nullptr, bool* lazy_out = nullptr, bool* unlit_out = nullptr) {
    const DScene&  s = c_scene;
    const uint32_t kind = c_view.v.integrator;
    const bool     mis = ZYGPU_INTEGRATOR_PTMIS == kind;
    const uint32_t ps = sampleSlot<POOL>(slot);
    PathHead       h;
    if (PRE) h = *pre;
    else loadHead<POOL>(P, slot, h);
    const uint32_t hit_prop = hit.prop;
    const bool     has_hit = kNull != hit_prop;
    // after Scene.intersect the probe's max_t is the hit distance (prop_tree.zig:76-79)
    Ray ray = Ray::init(xyz4(ro), xyz4(rd), 0.f, has_hit ? hit.t : ro.w);
    ray.time = pathTime<POOL, MESHL>(P, slot);
    const f3 wo = -ray.direction;

    if (!HEAVY) {  // anything that could make connectLight draw or evaluate goes to the heavy pass
        bool emissive = false;
        if (has_hit) {
            const ZygpuProp hp = s.props[hit_prop];
            const uint32_t  part = ZYGPU_SHAPE_TRIANGLE_MESH == hp.shape ? uint32_t(s.meshes[hp.mesh].triangle_parts[hit.primitive]) : 0u;
            emissive = 0 != (s.materials[s.material_ids[hp.parts_start + part]].flags & ZYGPU_MATF_EMISSIVE);
        }
        // infinite props: PathtracerMIS looks at them when the ray left the scene, the other two always walk the list
        const bool infinite = s.num_infinite_props > 0 && (ZYGPU_INTEGRATOR_PTMIS != kind || (!has_hit && kRayMaxT == ro.w));
        defer = connectsLights(h.flags) && (emissive || 0 != ucount || infinite);
        // a scene with shadow catchers (PathtracerMIS): a vertex on a catcher, or the one behind it, has the complete pass decide
        if (POOL && s.has_catcher && ZYGPU_INTEGRATOR_PTMIS == kind &&
            ((h.flags & ST_FROM_CATCHER) || (has_hit && (s.props[hit_prop].flags & ZYGPU_PROP_SHADOW_CATCHER))))
            defer = true;
    }
    // PRE == 1: the records generatePath<KEEP> left to this point (Vertex.init, vertex.zig:67-85: origin = the ray's, geo_n = 0,
    // the sums 0)
    const bool packed_prev = !POOL && 0 != P.packed_prev;
    auto commitFresh = [&]() {
        P.pool.ray_o[slot] = ro;
        P.pool.ray_d[slot] = rd;
        if (!packed_prev) {  // (packed: only the second pass wants them, and a vertex deferred to it gets them with its head below)
            P.pool.vorigin[slot] = f4(xyz4(ro), 0.f);
            P.pool.vgeo_n[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        clearAccumulators<true>(P, slot);
    };
    if (defer) {
        if (1 == PRE) {  // the second pass reads the path from the pool
            const PrevVertex camera{xyz4(ro), S3(0.f), 0.f};
            storeHead<POOL>(P, slot, h, packed_prev ? &camera : nullptr);  // (nothing has drawn since the camera sample's padding)
            P.pool.thr[slot] = make_float4(1.f, 1.f, 1.f, 0.f);
            commitFresh();
        }
        return;
    }
    // the fragment: needed here for emission (second pass) and by the POOL record; otherwise wf_scatter rebuilds it from
    // the hit (loadSurface) and this kernel only needs the hit's part
    Fragment frag;
    frag.prop = hit.prop;
    frag.part = 0;
    if (has_hit) {
        if (HEAVY || POOL) {
            makeFragment<MESHL, true>(s, ray, hit, frag);
        } else {
            const ZygpuProp hp = s.props[hit_prop];
            frag.part = ZYGPU_SHAPE_TRIANGLE_MESH == hp.shape ? uint32_t(s.meshes[hp.mesh].triangle_parts[hit.primitive]) : 0u;
        }
    }
    tm.mark(0);

    // ---- connectLight (pathtracer.zig:110-143, pathtracer_dl.zig:173-207, pathtracer_mis.zig:280-341)
    f3   energy = S3(0.f);
    bool shadow_catcher = false;                    // PathtracerMIS: the hit is on a shadow catcher
    f3   catcher_light[2] = {S3(0.f), S3(0.f)};     // this_light.occluded / .unoccluded
    if (HEAVY && connectsLights(h.flags)) {
        VertexIn v;
        if (packed_prev) {  // the words loadHead has just read as the (stale) Sobol buffer; see PrevVertex
            v.vorigin = mk3(h.smp.buffer[0], h.smp.buffer[1], h.smp.buffer[2]);
            v.vgeo_n = mk3(h.smp.buffer[3], h.smp.buffer[4], P.pool.indirect.get(slot).w);
        } else {
            v.vorigin = xyz4(P.pool.vorigin[slot]);
            v.vgeo_n = xyz4(P.pool.vgeo_n[slot]);
        }
        v.bxdf_pdf = rd.w;
        v.flags = h.flags;
        v.depth_surface = h.depth_surface;
        v.light_split_threshold = 0.f;
        // The three integrators look at the same three places in the same order (the hit, the unoccluding emitters the ray crossed, the
        // infinite props), each with its own conditions: one expansion of each place for all three (round 4: the complete pass of the
        // kernels that know every light kind held a dozen expansions of the light's pdf and was 477 KB of code)
        const bool is_pt = ZYGPU_INTEGRATOR_PT == kind, is_dl = ZYGPU_INTEGRATOR_PTDL == kind;
        if (is_pt) v.flags |= ST_SINGULAR;  // pathtracer.zig:115-117: no MIS
        if (!is_pt && !is_dl) v.light_split_threshold = splitThreshold(h.vdepth);
        bool at_hit = has_hit;
        if (is_dl && has_hit) {  // pathtracer_dl.zig:173-207: a light's own emission only where next-event estimation cannot have sampled it
            const uint32_t light_id = s.light_ids[s.props[frag.prop].parts_start + frag.part];
            at_hit = (h.flags & ST_SINGULAR) || kNull == light_id;
        }
        if (at_hit) energy = evaluateRadiance<MESHL, KINDS>(s, v, wo, frag, h.smp);
        tm.mark(1);
        if (!is_dl || (h.flags & ST_SINGULAR)) energy = energy + unoccludingEmission<MESHL, KINDS>(P, s, slot, vertexPlane<POOL>(), v, ray, h.smp);
        tm.mark(2);
        if (MESHL) {
            Ray  far = ray;
            f3*  catcher = nullptr;
            bool escaped = true, look;
            if (is_pt || is_dl) {
                look = true;
            } else if (POOL && s.has_catcher) {
                // pathtracer_mis.zig:296-337: on a shadow catcher, or behind one, the infinite props are looked at as if nothing had
                // been hit (the probe's max_t is RayMaxT for the loop)
                shadow_catcher = has_hit && 0 != (s.props[hit_prop].flags & ZYGPU_PROP_SHADOW_CATCHER);
                const bool handle = (h.flags & ST_FROM_CATCHER) || shadow_catcher;
                escaped = kRayMaxT == ray.max_t;
                look = handle || escaped;
                far.max_t = kRayMaxT;
                catcher = handle ? catcher_light : nullptr;
            } else {
                // RayMaxT == max_t (pathtracer_mis.zig:313): not for a ray that ended at its medium's bounds
                look = !has_hit && kRayMaxT == ray.max_t;
            }
            if (look) energy = energy + infiniteEmission<STATS, KINDS>(s, v, far, h.smp, is_dl, cnt, catcher, escaped);
        }
    }
    const float4 thr4 = PRE ? pre_thr : float4(P.pool.thr[slot]);
    f3           throughput = xyz4(thr4);
    const float  split_weight = POOL ? thr4.w : 1.f;
    Mediums      md;
    if (POOL) {
        if (0 == P.round) P.pool.pstate[ps] = P.pool.pstate.get(ps) & 255u;  // a new level: no children yet
        md.load(P, slot);
        // VolumeIntegrator.integrate for a medium that only absorbs (volume_integrator.zig:62-65, 105-128):
        // Beer-Lambert over the segment to the hit; Context.nextEvent runs before connectLight
        if (has_hit && md.index > 0) throughput = throughput * attenuation3(md.topAbsorption(), hit.t - 0.f);
    }
    nanTrace(P, energy, 8);
    nanTrace(P, throughput, 9);
    if (HEAVY && POOL && MESHL && mis && s.has_catcher) {
        // pathtracer_mis.zig:56-73.  (connectLight may have returned early for a specular vertex: then shadow_catcher is still the
        // hit's, decided here)
        shadow_catcher = has_hit && 0 != (s.props[hit_prop].flags & ZYGPU_PROP_SHADOW_CATCHER);
        const f3 split_throughput = throughput * S3(split_weight);
        if (shadow_catcher) {
            P.pool.sc_emis[slot] = f4(xyz4(P.pool.sc_emis[slot]) + split_throughput * catcher_light[1]);
            if (!(h.flags & ST_TRANSPARENT)) {  // `continue`: the vertex ends here, nothing of it reaches the result
                storeSamplers(P, ps, h);
                terminateVertex(P, ps, throughput, split_weight, h.flags, slot);
                return;
            }
        } else {
            P.pool.sc_occ[slot] = f4(xyz4(P.pool.sc_occ[slot]) + split_throughput * catcher_light[0]);
            P.pool.sc_unocc[slot] = f4(xyz4(P.pool.sc_unocc[slot]) + split_throughput * catcher_light[1]);
        }
    }
    // split_throughput = vertex.throughput * split_weight (pathtracer_mis.zig:61-62)
    ivalueAdd<!POOL>(P, ps, (POOL ? throughput * S3(split_weight) : throughput) * energy, h.depth_surface, mis ? 2u : 1u,
              0 == h.depth_surface, 0 != (h.flags & ST_SINGULAR));

    bool done = false;
    if (!has_hit || h.depth_surface >= c_view.v.max_depth_surface || 0 >= c_view.v.max_depth_volume) {
        if (mis && (h.flags & ST_TRANSPARENT)) throughput = throughput * (S3(1.f) - min3v(energy, S3(1.f)));
        done = true;
    } else {
        // russianRoulette (helper.zig:75-89); the draw happens whether or not it is needed
        const float r = h.smp.sample1D();
        const float continuation_probability = hmax3(throughput) / 0.1f;
        if (continuation_probability < 1.f) {
            if (r >= continuation_probability) done = true;
            else throughput = throughput / S3(continuation_probability);
        }
    }

    if (!done) {
        // ---- Vertex.sample (vertex.zig:137-181): the stochastic_r draw; the closure itself is rebuilt by wf_scatter /
        // wf_nee / wf_extend (loadSurface): from the hit, or in the POOL kernels from the record written here
        const uint32_t       mat_id = s.material_ids[s.props[frag.prop].parts_start + frag.part];
        const ZygpuMaterial& m = s.materials[mat_id];
        const float stochastic_r = h.smp.sample1D();  // rs.stochastic_r (vertex.zig:165)
        if (ZYGPU_MAT_LIGHT == m.kind) {
            // a pure emitter neither evaluates nor scatters (light_sample.zig:9-11, material_sample.zig:55-59):
            // no NEE draw, no bxdf sample, throughput = 0 (pathtracer_mis.zig:127-129)
            throughput = S3(0.f);
            done = true;
        } else {
            if (sr_out) *sr_out = stochastic_r;  // (the caller writes it with the hit, at the vertex' place in the alive queue)
            else P.pool.hit_sr[slot] = stochastic_r;
            // a closure that cannot be evaluated whatever the fragment: Glass whose roughness is the uniform 0 and stays 0 (no regularisation,
            // renderstate.zig:60-69): Sample.init's can_evaluate is false (glass_sample.zig:33-66), sampleLights draws and samples nothing
            if (POOL && unlit_out)
                *unlit_out = ZYGPU_MAT_GLASS == m.kind && 0.f == m.roughness && kNull == m.tex[ZYGPU_STEX_ROUGHNESS].image &&
                             (!mis || 0.f == c_view.v.regularize_roughness);
            if (material_out) *material_out = mat_id;
            // `lazy_out` (the first pass of a later tick, hit records by queue position): this pass has drawn twice (Russian roulette,
            // stochastic_r) and at most divided the throughput by the roulette's probability.  wf_scatter, which reads and rewrites the
            // path's line anyway, makes the two draws again from the line as it is (HitRecord::lazy), so this kernel writes neither the
            // sampler's 48 B nor the throughput back: two dirty sectors of the line less per vertex
            const bool lazy = !POOL && !HEAVY && 1 != PRE && nullptr != lazy_out;
            if (lazy_out) *lazy_out = lazy;
            if (!lazy) P.pool.thr[slot] = f4(throughput, POOL ? split_weight : thr4.w);  // (non-POOL: the spare lane may carry PrevVertex::alpha)
            if (POOL) {
                f3         geo_n = frag.geo_n, n = frag.n;
                const bool flipped = (m.flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo);
                if (flipped) {
                    geo_n = -geo_n;
                    n = -n;
                }
                P.pool.rec_p[slot] = f4(frag.p, __uint_as_float(mat_id));
                P.pool.rec_gn[slot] = f4(geo_n, flipped ? 1.f : 0.f);
                f3 color = S3(0.f);  // Substitute.sample's colour fetch (substitute_material.zig:120), for image maps
                if (MESHL && kNull != m.color_image && !(m.color_sampler & ZYGPU_TEX_PROCEDURAL)) color = sampleColorMapAt(s, m, frag.uv0, frag.uv1, stochastic_r, frag.p, geo_n, frag.trafo);
                P.pool.rec_n[slot] = f4(n, color.x);
                P.pool.rec_t[slot] = f4(frag.t, color.y);
                P.pool.rec_b[slot] = f4(frag.b, color.z);
                // rs.ior = Vertex.iorOutside (vertex.zig:87-92), rs.highest_priority (vertex.zig:178)
                const float ior_outside = frag.sameHemisphere(wo) ? md.topIor() : md.peekIor(frag.prop, frag.part);
                P.pool.rec_ior[slot] = make_float2(ior_outside, m.priority < md.highestPriority() ? 1.f : 0.f);
                if (MESHL && (m.flags & ZYGPU_MATF_EXTENDED)) P.pool.rec_uv[slot] = make_float2(frag.uv0, frag.uv1);
            }
            if (!lazy) storeHead<POOL>(P, slot, h);
            if (1 == PRE) commitFresh();
            alive = true;
        }
    }
    tm.mark(3);
    if (done) {
        if (POOL) {
            storeSamplers(P, ps, h);  // the sample's other vertices go on with these samplers
            terminateVertex(P, ps, throughput, split_weight, h.flags, slot);
        } else if (1 == PRE) {
            // a path that ends at its first vertex in this pass has gathered nothing (a hit with emission went to the second
            // pass): Sensor.addSample's colour is 0, and none of the path's records was ever written
            if (P.pixel_pcg) P.pixel_pcg[slot] = make_uint2(uint32_t(h.smp.rng.state), uint32_t(h.smp.rng.state >> 32));
            const float alpha = (h.flags & ST_TRANSPARENT) ? mx(1.f - average3(throughput), 0.f) : 1.f;
            P.stage_color[slot] = make_float4(0.f, 0.f, 0.f, alpha);
        } else {
            finalize(P, slot, throughput, h.flags, h.smp.rng);
        }
    }
    tm.mark(4);
}

struct HitRecord {  // what wf_scatter's closure set-up reads of the vertex' hit: by slot from the pool, or by queue position
    uint4    hit;       // t, u, v (bits), primitive
    uint32_t prop, ucount;
    float    stochastic_r;
    // the hit part's material, which the shading pass has looked up (prop -> part -> material: three dependent loads that
    // loadSurface then does not repeat), or kNull: not recorded (the records by slot have no word for it)
    uint32_t material;
    // the first shading pass left the path's line alone (shadeVertex `lazy_out`): wf_scatter makes its two draws again
    bool     lazy;
};
ZD HitRecord loadHitRecord(const Params& P, uint32_t slot) {
    return {P.pool.hit[slot], P.pool.hit_prop[slot], P.pool.ucount[slot], P.pool.hit_sr[slot], kNull, false};
}
ZD HitRecord loadQueuedHit(const Params& P, uint32_t position) {
    const uint4 a = P.hit_q[2u * size_t(position)], b = P.hit_q[2u * size_t(position) + 1u];
    return {a, b.x, b.y & 0x7FFFFFFFu, __uint_as_float(b.z), b.w, 0 != (b.y >> 31)};
}
ZD void storeQueuedHit(const Params& P, uint32_t position, const Hit& hit, uint32_t ucount, float stochastic_r, uint32_t material, bool lazy) {
    P.hit_q[2u * size_t(position)] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
    P.hit_q[2u * size_t(position) + 1u] = make_uint4(hit.prop, ucount | (lazy ? 0x80000000u : 0u), __float_as_uint(stochastic_r), material);
}

// A block of 256 threads deals its alive-queue entries to its threads prop by prop (a counting sort by the hit prop's low bits in LDS:
// the vertices of a prop share their materials and, on a flat one, their side of every light), so that a wave works on one or two
// surfaces and not on every surface of the scene, as it does past the first tick.  Which thread works on which vertex changes nothing in
// what is stored for the vertex; the order of what the block appends to a queue changes, which nothing reads.  Whole blocks only.
template <bool POOL>
ZD void dealByProp(const Params& P, uint32_t& at, bool& has_entry) {
    __shared__ uint32_t bucket[16];
    __shared__ uint32_t entry[256];
    if (threadIdx.x < 16u) bucket[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t key = 0, position = 0;
    if (has_entry) {
        key = ((!POOL && P.hit_q) ? P.hit_q[2u * size_t(at) + 1u].x : P.pool.hit_prop[P.alive_in[at]]) & 15u;
        position = atomicAdd(&bucket[key], 1u);
    }
    __syncthreads();
    uint32_t first = 0, total = 0;
    for (uint32_t k = 0; k < 16u; ++k) {
        const uint32_t n = bucket[k];
        first += k < key ? n : 0u;
        total += n;
    }
    if (has_entry) entry[first + position] = at;
    __syncthreads();
    has_entry = threadIdx.x < total;
    at = has_entry ? entry[threadIdx.x] : 0u;
}

// Two passes share shadeVertex.  HEAVY = false walks the extension queue and keeps the common case: a vertex whose hit is not
// emissive and whose ray crossed no unoccluding light has no emission to evaluate (no sampler draw, no light-tree pdf,
// no SphQuad), so that instantiation is small enough for twice the occupancy; the others go to the `heavy` queue, which
// the HEAVY = true pass processes with the complete connectLight.
// POOL: `slot` below is a vertex index
template <bool STATS, bool MESHL, bool HEAVY, bool POOL, uint32_t KINDS = kAllLightKinds>
__global__ void __launch_bounds__(256) ZYG_WAVES(HEAVY ? (MESHL ? ZYG_SHADE_BIG_WAVES : ZYG_SHADE_WAVES) : ZYG_SHADE_LIGHT_WAVES) wf_shade(Params P) {
    const uint32_t cur = P.parity;
    Counters       cnt{0, 0, 0};
    const QueueCounts* counts = HEAVY ? &P.ctl->heavy_count[cur] : extCount<POOL>(P, cur, P.round);
    // one queue entry: shade it, append the path to the queues it goes on in
    auto entry = [&](uint32_t i, uint32_t at, uint32_t total) {
        SectionTimer tm;
        tm.start();
        bool       alive = false, defer = false;
        uint32_t   slot = 0, alive_key = 0, ucount = 0, position = 0;
        Hit        hit{0.f, 0.f, 0.f, 0u, kNull};
        float      stochastic_r = 0.f;
        uint32_t   material = kNull;
        bool       lazy = false, unlit = false;
        const bool queued = !POOL && nullptr != P.hit_q;
        if (i < total) {
            slot = HEAVY ? P.q.heavy[at] : extQueue<POOL>(P, cur, P.round)[at];
            const uint4 hr = P.pool.hit[slot];
            hit = Hit{__uint_as_float(hr.x), __uint_as_float(hr.y), __uint_as_float(hr.z), hr.w, P.pool.hit_prop[slot]};
            alive_key = hit.prop;
            ucount = P.pool.ucount[slot];
            shadeVertex<STATS, MESHL, HEAVY, POOL, 0, KINDS>(P, slot, P.pool.ray_o[slot], P.pool.ray_d[slot], hit, ucount, alive, defer, cnt, tm, nullptr,
                                                             make_float4(1.f, 1.f, 1.f, 0.f), queued ? &stochastic_r : nullptr, queued ? &material : nullptr,
                                                             (queued && !HEAVY && P.lazy_head) ? &lazy : nullptr, POOL ? &unlit : nullptr);
        }
        if (HEAVY && !POOL && P.regen_sample) {
            // the regenerating kernels: no queue behind this pass; a path that ended here leaves its slot to wf_rscatter
            if (i < total && !alive) P.regen_sample[slot] = kNull;
        } else {
            if (!HEAVY) {
                queuePush(P.q.heavy, &P.ctl->heavy_count[cur], P.queue_capacity, defer, slot);
                __syncthreads();  // queuePush's shared words are reused below
            }
            if (!POOL && P.alive_keys > 1) alivePushKeyed<4>(P, cur, false, alive, alive_key & (kAliveKeys - 1u), slot);
            else queuePush(P.alive_out, &P.ctl->alive_count[cur][0], P.queue_capacity, alive, slot, POOL && unlit && 0 != P.alive_ends, &position);
            if (queued && alive) storeQueuedHit(P, position, hit, ucount, stochastic_r, material, lazy);
        }
        tm.mark(5);
        tm.flush(P.ctl, 0);
    };
    if (HEAVY) {
        // The second pass has few entries (vertices with emission to evaluate) and used to be launched like the others, one
        // block per 256 slots of the tick, nearly all of them leaving at once (configs[1]: 1.30 -> 1.15 ms of a 43 ms step).
        // Now a grid of a few blocks per CU walks the queue with the grid's stride (zygpu.hip heavyGrid).
        uint32_t       at;
        const uint32_t total = queueOpen(counts, P.queue_capacity, 0u, at);
        const uint32_t rounded = (total + 255u) & ~255u;  // whole blocks take part in queuePush
        for (uint32_t base = blockIdx.x * blockDim.x; base < rounded; base += gridDim.x * blockDim.x) {
            const uint32_t i = base + threadIdx.x;
            queueOpen(counts, P.queue_capacity, i, at);
            entry(i, at, total);
            __syncthreads();  // queuePush's shared words are reused by the next round
        }
    } else {
        if (blockIdx.x * blockDim.x >= P.ctl->bound[cur]) return;
        const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
        uint32_t       at;
        const uint32_t total = queueOpen(counts, P.queue_capacity, i, at);
        const uint32_t rounded = (total + 255u) & ~255u;  // whole blocks take part in queuePush
        if (i < rounded) entry(i, at, total);
    }
    if (STATS) flushCounters(P, 1, cnt, 0);
}

// The same for the POOL scenes' complete instantiations (wf_intersect<false, true, true, WIDEC> + wf_shade<false, true, false, true>; option
// "fuse_pool"): the vertex' first line is read once, the hit record goes to its second line together with the fragment's records (one
// full line written, no 32 B write and no read of that line between the two kernels).  The rounds of a tick share the alive / heavy
// counters, so they are reset by a launch of their own in front (wf_pool_reset) instead of by this kernel's first block.
#ifndef ZYG_TRACE_SHADE_POOL_WAVES
#define ZYG_TRACE_SHADE_POOL_WAVES 3
#endif
template <bool POOL>  // (a template so that only the unit that instantiates it emits it)
__global__ void __launch_bounds__(256) wf_pool_reset(Params P) {  // one block
    if (threadIdx.x >= 64) return;
    const uint32_t cur = P.parity;
    if (0 == P.round)
        for (uint32_t k = 0; k < kPoolVertices; ++k) queueReset(&P.ctl->ext_pool_count[cur ^ 1u][k]);
    aliveReset(P.ctl->alive_count[cur]);
    queueReset(&P.ctl->shadow_count[0]);
    queueReset(&P.ctl->shadow_count[1]);
    queueReset(&P.ctl->heavy_count[cur]);
}

// FIRST (round 5): the batch's first tick generates its camera samples itself (wf_generate<POOL>'s body: entry i is slot i, `first_paths` of
// them) right in front of their trace: the sample's two lines are still in the caches when the trace reads them, and wf_generate's launch
// and the round trip through the extension queue are gone (the outdoor scene at 3840x2160: wf_generate was 6 % of the frame).
template <bool WIDEC, bool FIRST = false>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_TRACE_SHADE_POOL_WAVES) wf_trace_shade_pool(Params P, uint32_t first_paths) {
    const uint32_t cur = P.parity;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at = i;
    const uint32_t n = FIRST ? first_paths : queueOpen(extCount<true>(P, cur, P.round), P.queue_capacity, i, at);
    if (0 == i) P.ctl->bound[cur] = n;
    if (blockIdx.x * blockDim.x >= n) return;  // (whole blocks take part in queuePush)
    Counters     cnt{0, 0, 0};
    SectionTimer tm;
    tm.start();
    bool     alive = false, defer = false, unlit = false;
    uint32_t slot = 0;
    if (i < n) {
        if (FIRST) {
            slot = i;
            generatePath<true, true>(P, slot);
        } else {
            slot = extQueue<true>(P, cur, P.round)[at];  // a vertex index
        }
        const float4   o = P.pool.ray_o[slot], d = P.pool.ray_d[slot];
        const uint32_t meta = P.pool.vmeta.get(slot);
        const uint32_t depth = (meta >> 3) & 31u, flags = (meta >> 13) & 63u;
        Ray            ray = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
        ray.time = pathTime<true, true>(P, slot);
        float4 ray_o = o;
        {
            // inside a dielectric the ray ends where it leaves the medium prop's bounding box (wf_intersect)
            Mediums md;
            md.load(P, slot);
            if (md.index > 0) {
                const uint32_t mp = md.prop(md.index - 1);
                const float    limit = aabbIntersectP(c_scene.aabbs[2 * mp], c_scene.aabbs[2 * mp + 1], ray);
                ray.max_t = mn(offsetF(limit), o.w);
                ray_o = f4(ray.origin, ray.max_t);
                P.pool.ray_o[slot] = ray_o;  // (the complete pass and the scatter kernels read the shortened ray)
            }
        }
        Hit                                  hit;
        typename MaskDraw<true, true>::type draw(P, slot);
        sceneIntersect<false, true, WIDEC>(c_scene, ray, depth, hit, cnt, draw);
        draw.flush();
        if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
        // after Scene.intersect the probe's max_t is the hit distance (prop_tree.zig:76-79)
        const uint32_t ucount = walksUnoccluding(flags) ? unoccludingCandidates<false, true>(P, c_scene, slot, vertexPlane<true>(), depth, ray, cnt) : 0u;
        shadeVertex<false, true, false, true, 0, kAllLightKinds>(P, slot, ray_o, d, hit, ucount, alive, defer, cnt, tm, nullptr, make_float4(1.f, 1.f, 1.f, 0.f),
                                                                 nullptr, nullptr, nullptr, &unlit);
        if (alive || defer) {  // the complete pass, wf_light and the closure half read the hit
            P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
            P.pool.hit_prop[slot] = hit.prop;
            P.pool.ucount[slot] = ucount;
        }
    }
    queuePush(P.q.heavy, &P.ctl->heavy_count[cur], P.queue_capacity, defer, slot);
    __syncthreads();  // queuePush's shared words are reused below
    queuePush(P.alive_out, &P.ctl->alive_count[cur][0], P.queue_capacity, alive, slot, unlit && 0 != P.alive_ends);
}

// wf_intersect and wf_shade's first pass in one kernel (non-POOL): the ray, the hit and the candidate count stay in
// registers, so the path's line is read once instead of twice and the 32 B hit record is only written for the few vertices
// the second pass will look at.  The alive / heavy queue counters are per tick parity here: this kernel appends to the
// tick's and resets the next tick's (wf_intersect, a kernel of its own, could reset the ones wf_shade was about to fill).
// FIRST: the batch's first tick generates its camera rays itself (wf_generate's body) instead of reading them back from the
// pool through the extension queue: entry i is slot i, `first_paths` of them.
// UTLAS: the solid and unoccluding props are walked as the scene's tables (dscene.hpp TlasEntry; not with STATS or MESHL)
// WIDE (with UTLAS, relaxed units): meshes are walked through their four-wide re-layout (bvh4.hpp)
template <bool STATS, bool MESHL, bool FIRST, bool UTLAS = false, bool WIDE = false>
__global__ void __launch_bounds__((traceShadeBlock<UTLAS, WIDE>())) ZYG_WAVES(ZYG_TRACE_SHADE_WAVES) wf_trace_shade(Params P, uint32_t first_paths) {
    constexpr uint32_t kTraceShadeBlock = traceShadeBlock<UTLAS, WIDE>();
    static_assert(!UTLAS || (!STATS && !MESHL), "the prop tables hold Rectangles and meshes only and count nothing");
    const uint32_t cur = P.parity;
    if (!FIRST && blockIdx.x * blockDim.x >= P.ctl->bound[cur ^ 1u]) return;  // the previous tick's queue was no longer
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at = i;
    const uint32_t n = FIRST ? first_paths : queueOpen(&P.ctl->ext_count[cur], P.queue_capacity, i, at);
    if (0 == i) P.ctl->bound[cur] = n;
    if (0 == blockIdx.x && threadIdx.x < 64) {
        queueReset(&P.ctl->ext_count[cur ^ 1u]);  // this tick's wf_scatter fills it
        aliveReset(P.ctl->alive_count[cur ^ 1u]);
        queueReset(&P.ctl->heavy_count[cur ^ 1u]);
        queueReset(&P.ctl->shadow_count[0]);  // staged NEE: this tick's wf_nee rounds
        queueReset(&P.ctl->shadow_count[1]);
        if (0 == threadIdx.x) P.ctl->mesh_fetch = 0;  // the next tick's wf_mesh_closest
    }
    Counters       cnt{0, 0, 0}, shade_cnt{0, 0, 0};
    const uint32_t rounded = (n + (kTraceShadeBlock - 1u)) & ~(kTraceShadeBlock - 1u);  // whole blocks take part in queuePush
    const DScene& scene = c_scene;
    if (i < rounded) {
        SectionTimer tm;
        tm.start();
        bool       alive = false, defer = false;
        uint32_t   slot = 0, alive_key = 0, queued_ucount = 0, position = 0, queued_material = kNull;
        Hit        queued_hit{0.f, 0.f, 0.f, 0u, kNull};
        float      stochastic_r = 0.f;
        bool       queued_lazy = false;
        const bool queued = nullptr != P.hit_q;
#ifndef ZYG_SORT_BLOCK
#define ZYG_SORT_BLOCK 0
#endif
        uint32_t sorted_slot = kNull;
        if (WIDE && !FIRST && ZYG_SORT_BLOCK) {
            // -DZYG_SORT_BLOCK=1 (experiment): the block's 256 extension rays are dealt to its threads in the order of a 6-bit key: the
            // octant of the cell (2 x 2 x 2 over the first large mesh's bounds) in which the ray enters those bounds, and the octant of
            // its direction, so that a wave's rays start their walk in one region and visit a node's children in one order.
            __shared__ uint32_t bins[64];
            __shared__ uint32_t dealt[kTraceShadeBlock];
            const bool valid = i < n;
            uint32_t   mine = kNull, key = 63u;
            if (valid) {
                mine = P.q.ext[cur][at];
                const float4 o = P.pool.ray_o[mine], d = P.pool.ray_d[mine];
                const f3     lo = xyz4(scene.heavy_aabbs[0]), hi = xyz4(scene.heavy_aabbs[1]);
                const f3     inv = S3(1.f) / xyz4(d);
                const f3     t0 = (lo - xyz4(o)) * inv, t1 = (hi - xyz4(o)) * inv;
                const float  enter = mx(mx(mn(t0.x, t1.x), mn(t0.y, t1.y)), mx(mn(t0.z, t1.z), 0.f));
                const f3     p = (xyz4(o) + S3(enter) * xyz4(d) - lo) / (hi - lo);
                const uint32_t cell = (p.x > 0.5f ? 1u : 0u) | (p.y > 0.5f ? 2u : 0u) | (p.z > 0.5f ? 4u : 0u);
                const uint32_t octant = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
                key = cell * 8u + octant;
            }
            if (threadIdx.x < 64u) bins[threadIdx.x] = 0u;
            __syncthreads();
            const uint32_t rank = valid ? atomicAdd(&bins[key], 1u) : 0u;
            __syncthreads();
            if (threadIdx.x < 64u) {  // exclusive prefix over the 64 bins by the block's first wave
                const uint32_t count = bins[threadIdx.x];
                uint32_t       incl = count;
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o);
                    if (int(threadIdx.x) >= o) incl += up;
                }
                bins[threadIdx.x] = incl - count;
            }
            __syncthreads();
            if (valid) dealt[bins[key] + rank] = mine;
            __syncthreads();
            if (valid) sorted_slot = dealt[threadIdx.x];  // (a block's valid entries are its first ones)
        }
        if (i < n) {
            slot = FIRST ? i : (WIDE && ZYG_SORT_BLOCK) ? sorted_slot : P.q.ext[cur][at];
            // the fresh path stays in registers until it is shaded, unless the traversal may draw from its sampler through the
            // pool (opacity masks: the MESHL instantiations)
            constexpr bool kFresh = FIRST && !MESHL;
            // later ticks read the head and the throughput with the ray, before the traversal, instead of behind it (configs[1]:
            // wf_trace_shade -5 %); not where the traversal may draw from the sampler through the pool (masks: MESHL)
            constexpr int kPre = MESHL ? 0 : (FIRST ? 1 : 2);
            PathHead       fresh;
            float4         fresh_thr = make_float4(1.f, 1.f, 1.f, 0.f);
            float4         fresh_o, fresh_d;
            if (FIRST) generatePath<false, MESHL, kFresh>(P, slot, &fresh, &fresh_o, &fresh_d);
            if (2 == kPre) {
                loadHead(P, slot, fresh);
                fresh_thr = P.pool.thr[slot];
            }
            tm.mark(8);
            const float4   o = kFresh ? fresh_o : float4(P.pool.ray_o[slot]), d = kFresh ? fresh_d : float4(P.pool.ray_d[slot]);
            const uint32_t meta = FIRST ? uint32_t((ST_PRIMARY | ST_TRANSPARENT | ST_SINGULAR) << 13) : P.pool.q0.get(slot).y;
            const uint32_t depth = (meta >> 3) & 31u, flags = (meta >> 13) & 63u;
            Ray            ray = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
            ray.time = pathTime<false, MESHL>(P, slot);
            Hit            hit;
            if (WIDE && !FIRST && P.mesh_pass && i >= P.ctl->ext_count[cur].n[0]) {
                // a back entry: wf_mesh_closest has its complete closest hit
                const uint4 h = P.pool.hit[slot];
                hit.prop = P.pool.hit_prop[slot];
                hit.t = __uint_as_float(h.x);
                hit.u = __uint_as_float(h.y);
                hit.v = __uint_as_float(h.z);
                hit.primitive = h.w;
                if (kNull != hit.prop) ray.max_t = hit.t;
                // (wf_mesh_closest8 walks the large meshes only: the Rectangles and small meshes are tested here, by every lane together)
                if (UTLAS && 2u == P.mesh_pass) tableClosestBeforeLargeMeshes(scene, ray, depth, hit);
            } else if (UTLAS) {
#if ZYG_FAST
                sceneIntersectUniform<false, WIDE>(scene, ray, depth, hit);
#else
                sceneIntersectUniform<true, false>(scene, ray, depth, hit);
#endif
            } else {
                typename MaskDraw<false, MESHL>::type draw(P, slot);
                sceneIntersect<STATS, MESHL>(scene, ray, depth, hit, cnt, draw);
                draw.flush();
            }
            tm.mark(9);
            if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
            alive_key = hit.prop;
            const uint32_t ucount = walksUnoccluding(flags) ? unoccludingCandidates<STATS, MESHL, UTLAS>(P, scene, slot, vertexPlane<false>(), depth, ray, cnt) : 0u;
            tm.mark(10);
            shadeVertex<STATS, MESHL, false, false, kPre>(P, slot, o, d, hit, ucount, alive, defer, shade_cnt, tm, &fresh, fresh_thr,
                                                          queued ? &stochastic_r : nullptr, queued ? &queued_material : nullptr,
                                                          (queued && !FIRST && P.lazy_head) ? &queued_lazy : nullptr);
            // wf_shade's second pass and wf_scatter's closure set-up read the hit from the pool: by slot, or (the vertices that go
            // straight on to wf_scatter, Params::hit_q) at their place in the alive queue, behind the append below
            if (defer || (alive && !queued)) {
                P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
                P.pool.hit_prop[slot] = hit.prop;
                P.pool.ucount[slot] = ucount;
            }
            queued_hit = hit;
            queued_ucount = ucount;
        }
        // one pair of barriers for both appends (a deferred vertex is appended to the alive queue by wf_shade's second pass):
        // configs[1] wf_trace_shade 126.0 -> 123.9 ms per 8 steps against two queuePush calls
        if (UTLAS && P.alive_keys > 1) alivePushKeyed<kTraceShadeBlock / 64>(P, cur, defer, alive && !defer, alive_key & (kAliveKeys - 1u), slot);
        else queuePushTwo<kTraceShadeBlock / 64>(P.q.heavy, &P.ctl->heavy_count[cur], defer, P.alive_out, &P.ctl->alive_count[cur][0], alive && !defer, slot, &position);
        if (queued && alive && !defer) storeQueuedHit(P, position, queued_hit, queued_ucount, stochastic_r, queued_material, queued_lazy);
        tm.mark(11);
        tm.flush(P.ctl, 0);
    }
    if (STATS) {
        flushCounters(P, 0, cnt, (0 == blockIdx.x && 0 == threadIdx.x) ? n : 0);
        flushCounters(P, 1, shade_cnt, 0);
    }
}

// ---------------------------------------------------------------- closure from the hit record

struct Surface {
    f3             p;
    f3             frag_geo_n;  // the fragment's own geometric normal (before the two-sided flip), for Fragment.offsetP
    uint32_t       material;    // scene.propMaterialId(frag.prop, frag.part)
    float          parent_pdf;  // POOL: vertex.bxdf_pdf as the ray carried it (ray_d.w), for the children of a .Straight event
    MaterialSample ms;
};

// rs.reg_weight / rs.reg_alpha (vertex.zig:168-169): PathtracerMIS' regularize_roughness and the roughness of the path's last
// scattering event, which travels in the spare lane of vertex.geo_n (read only when the setting is on)
ZD Regularize loadRegularize(const Params& P, uint32_t slot) {
    Regularize   r;
    const float  w = ZYGPU_INTEGRATOR_PTMIS == c_view.v.integrator ? c_view.v.regularize_roughness : 0.f;
    if (0.f != w) {
        r.weight = w;
        r.alpha = P.packed_prev ? P.pool.thr.get(slot).w : P.pool.vgeo_n.get(slot).w;
    }
    return r;
}

ZD float* texBlock(const Params& P, uint32_t slot) { return P.tex_values + size_t(slot) * kTexPlanes; }
// a procedural colour as wf_textures left it
ZD f3 textureColor(const Params& P, uint32_t slot) {
    const float* t = texBlock(P, slot);
    return mk3(t[kTexColor], t[kTexColor + 1u], t[kTexColor + 2u]);
}
// The sampler dimensions the vertex' procedural textures drew in wf_textures (a Mix picks one of its two children with one,
// procedural_mix.zig:17-26): Material.sample comes before the vertex' light and bxdf samples (pathtracer_mis.zig:93), so the kernel
// that draws those skips as many first
ZD uint32_t skipTextureDraws(const Params& P, uint32_t slot, uint32_t material, PathHead& h) {
    if (!P.tex_values || !(c_scene.materials[material].flags & ZYGPU_MATF_PROCEDURAL)) return 0u;
    const uint32_t n = __float_as_uint(texBlock(P, slot)[kTexDraws]);
    for (uint32_t k = 0; k < n; ++k) (void)h.smp.sample1D();
    return n;
}

// POOL: from the record wf_shade wrote.  Otherwise from the hit: the fragment is rebuilt here (Shape.fragment), followed by
// Vertex.sample's two-sided flip (vertex.zig:155-161) and the colour-map fetch with the stochastic_r wf_shade drew: the path
// carries 36 B of hit instead of an 80 B record.  EXTRA: Sphere / Disk / Cube fragments.
template <bool POOL = false, bool EXTRA = false>
ZD Surface loadSurface(const Params& P, uint32_t slot, uint32_t flags, const HitRecord* record = nullptr) {
    if (!POOL) {
        const DScene&  s = c_scene;
        const float4   ro = P.pool.ray_o[slot], rd = P.pool.ray_d[slot];
        const uint4    hr = record ? record->hit : uint4(P.pool.hit[slot]);
        const Hit      hit{__uint_as_float(hr.x), __uint_as_float(hr.y), __uint_as_float(hr.z), hr.w, record ? record->prop : uint32_t(P.pool.hit_prop[slot])};
        const float    hit_sr = (EXTRA) ? (record ? record->stochastic_r : float(P.pool.hit_sr[slot])) : 0.f;
        Ray            ray = Ray::init(xyz4(ro), xyz4(rd), 0.f, hit.t);
        ray.time = pathTime<false, EXTRA>(P, slot);
        Fragment       frag;
        makeFragment<EXTRA>(s, ray, hit, frag);
        const f3             wo = -ray.direction;
#ifndef ZYG_EXPERIMENT_MATERIAL_CHAIN
#define ZYG_EXPERIMENT_MATERIAL_CHAIN 0  // 1: the A/B of HitRecord::material (the look-up is always repeated)
#endif
        const uint32_t       material = (!ZYG_EXPERIMENT_MATERIAL_CHAIN && record && kNull != record->material)
                                            ? record->material
                                            : uint32_t(s.material_ids[s.props[frag.prop].parts_start + frag.part]);
        const ZygpuMaterial& m = s.materials[material];
        f3                   geo_n = frag.geo_n, n = frag.n;
        if ((m.flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo)) {
            geo_n = -geo_n;
            n = -n;
        }
        f3 color = mk3(m.color_a[0], m.color_a[1], m.color_a[2]);
        if (EXTRA && kNull != m.color_image)
            color = (m.color_sampler & ZYGPU_TEX_PROCEDURAL) ? textureColor(P, slot) : sampleColorMapAt(s, m, frag.uv0, frag.uv1, hit_sr, frag.p, geo_n, frag.trafo);
        const bool caustics = (flags & ST_PRIMARY) ? true : 0 != c_view.v.caustics_path;  // causticsResolve
        Surface    sf;
        sf.p = frag.p;
        sf.parent_pdf = 0.f;  // POOL only
        sf.frag_geo_n = frag.geo_n;
        sf.material = material;
        const Regularize reg = loadRegularize(P, slot);
        if (EXTRA && (m.flags & ZYGPU_MATF_EXTENDED)) {
            const FragmentExtra fx{frag.uv0, frag.uv1, hit_sr, frag.p, frag.trafo, P.tex_values ? texBlock(P, slot) : nullptr};
            sf.ms = makeMaterialSample<EXTRA>(s, m, wo, geo_n, frag.t, frag.b, n, caustics, color, 1.f, false, &fx, reg);
        } else {
            sf.ms = makeMaterialSample(s, m, wo, geo_n, frag.t, frag.b, n, caustics, color, 1.f, false, nullptr, reg);
        }
        return sf;
    }
    const float4 rp = P.pool.rec_p[slot];
    const float4 rg = P.pool.rec_gn[slot];
    const float4 rn = P.pool.rec_n[slot], rt = P.pool.rec_t[slot], rb = P.pool.rec_b[slot];
    const f3     geo_n = xyz4(rg), n = xyz4(rn);
    const f3     t = xyz4(rt), b = xyz4(rb);
    const float4 prd = P.pool.ray_d[slot];
    const f3     wo = -xyz4(prd);
    const ZygpuMaterial& m = c_scene.materials[__float_as_uint(rp.w)];
    // a colour map was sampled by wf_shade (it has the uv and the stochastic_r draw) into the records' spare lanes
    const f3     color = kNull != m.color_image ? ((EXTRA && (m.color_sampler & ZYGPU_TEX_PROCEDURAL)) ? textureColor(P, slot) : mk3(rn.w, rt.w, rb.w))
                                                : mk3(m.color_a[0], m.color_a[1], m.color_a[2]);
    const bool   caustics = (flags & ST_PRIMARY) ? true : 0 != c_view.v.caustics_path;  // causticsResolve
    Surface      sf;
    sf.p = xyz4(rp);
    sf.parent_pdf = prd.w;
    sf.frag_geo_n = 0.f != rg.w ? -geo_n : geo_n;
    sf.material = __float_as_uint(rp.w);
    const Regularize reg = loadRegularize(P, slot);
    if (POOL) {
        const float2 ri = P.pool.rec_ior[slot];
        if (EXTRA && (m.flags & ZYGPU_MATF_EXTENDED)) {
            const float2        uv = P.pool.rec_uv[slot];
            const FragmentExtra fx{uv.x, uv.y, P.pool.hit_sr[slot], sf.p,
                                   propTrafoAt<true>(c_scene, P.pool.hit_prop[slot], pathTime<true, true>(P, slot)),
                                   P.tex_values ? texBlock(P, slot) : nullptr};
            sf.ms = makeMaterialSample<EXTRA>(c_scene, m, wo, geo_n, t, b, n, caustics, color, ri.x, 0.f != ri.y, &fx, reg);
        } else {
            sf.ms = makeMaterialSample(c_scene, m, wo, geo_n, t, b, n, caustics, color, ri.x, 0.f != ri.y, nullptr, reg);
        }
    } else {
        sf.ms = makeMaterialSample(c_scene, m, wo, geo_n, t, b, n, caustics, color, 1.f, false, nullptr, reg);
    }
    return sf;
}

// Worker.commonAOV (worker.zig:209-241) into the sample's staged aov.Value; POOL: the sample's vertices in their order
ZD void commonAov(const Params& P, uint32_t slot, uint32_t ps, const PathHead& h, const Surface& sf, f3 throughput,
                  const HitRecord* record = nullptr) {
    const MaterialSample& ms = sf.ms;
    const uint32_t        slots = c_view.v.aov_slots;
    float4*               av = P.stage_aov + ps;
    if ((h.flags & ST_PRIMARY) && (slots & (1u << ZYGPU_AOV_ALBEDO)) && ms.can_evaluate) {
        // MaterialSample.aovAlbedo (material_sample.zig:38-45, substitute_sample.zig:80-86)
        const f3 a = ZYGPU_MAT_SUBSTITUTE == ms.kind ? lerp3(ms.albedo, ms.f0, S3(ms.metallic))
                     : ZYGPU_MAT_DEBUG == ms.kind    ? ms.albedo
                     : ZYGPU_MAT_GLASS == ms.kind    ? S3(1.f)
                                                     : S3(0.f);
        av[ZYGPU_AOV_ALBEDO * P.aov_plane] = f4(throughput * a);
    }
    if (0 == h.depth_surface) {
        if (slots & (1u << ZYGPU_AOV_GEOMETRIC_NORMAL)) av[ZYGPU_AOV_GEOMETRIC_NORMAL * P.aov_plane] = f4(ms.geo_n);
        if (slots & (1u << ZYGPU_AOV_SHADING_NORMAL)) av[ZYGPU_AOV_SHADING_NORMAL * P.aov_plane] = f4(ms.frame.z);
        if (slots & (1u << ZYGPU_AOV_ROUGHNESS)) av[ZYGPU_AOV_ROUGHNESS * P.aov_plane].x = sqrtf(ms.a0);
        if (slots & (1u << ZYGPU_AOV_DEPTH)) av[ZYGPU_AOV_DEPTH * P.aov_plane].x = __uint_as_float(record ? record->hit.x : P.pool.hit.get(slot).x);
        if (slots & (1u << ZYGPU_AOV_MATERIAL_ID)) av[ZYGPU_AOV_MATERIAL_ID * P.aov_plane].x = float(1u + sf.material);
    }
}

// Fragment.offsetP (intersection.zig:115-119; uvw[3], the offset, is 0 for rectangles and meshes)
ZD f3 surfaceOffsetP(f3 p, f3 geo_n, f3 v) {
    const f3 nn = dot3(geo_n, v) > 0.f ? geo_n : -geo_n;
    return offsetRay(fma3(S3(0.f), nn, p), nn);
}

// ---------------------------------------------------------------- next-event estimation, staged
// sampleLights (pathtracer_mis.zig:174-212, pathtracer_dl.zig:100-137) picks the lights first (light tree, one sampler
// draw), then handles them one after the other: evaluateLight draws the light sample (2-D), tests the shadow ray and,
// only if it is unoccluded, draws one more dimension (light/light.zig:127).  That draw makes the picks of a vertex
// sequential, so they run as rounds:  wf_nee(r) finishes pick r-1 (adds its contribution if wf_shadow found the ray
// unoccluded, performs the draw) and starts pick r (light sample, shadow ray + the pre-weighted contribution into the
// pool, slot into the shadow queue);  wf_shadow traces the queue with every lane busy;  wf_extend finishes the last
// pick, then samples the bxdf.  Radiance, bsdf value and MIS weight of a pick do not depend on the visibility test, so
// they are computed when the pick starts and the result is bit-identical to the reference's order.

constexpr uint32_t kNeeCountMask = 127u;  // PathHead::nee [0:6] number of picks of this vertex
constexpr uint32_t kNeePending = 128u;    //               [7]   the previous round issued a shadow ray


// evaluateLight up to the shadow test (pathtracer_mis.zig:214-278 / pathtracer_dl.zig:139-167); true = shadow ray issued
template <bool SUBST>
ZD bool startLightPick(const Params& P, uint32_t slot, PathHead& h, const Surface& sf, LightPick pick, bool mis) {
    const DScene&         s = c_scene;
    const MaterialSample& ms = sf.ms;
    const ZygpuLight      light = s.lights[pick.offset];
    const Trafo           trafo = loadTrafo(s, light.prop);
    const ZygpuMaterial&  lm = s.materials[s.material_ids[s.props[light.prop].parts_start + light.part]];
    SampleTo              ls;
    if (!rectangleSampleTo(sf.p, ms.geo_n, trafo, 0 != light.two_sided, ms.translucent, h.smp, ls)) return false;
    const f3 sorigin = surfaceOffsetP(sf.p, sf.frag_geo_n, ls.wi);
    f3       sdir;
    float    slen;
    shadowRay(sorigin, ls, sdir, slen);
    const f3         radiance = materialRadiance(s, lm, ls.wi, trafo, light.prop, light.part, false);
    const BxdfResult bx = ms.template evaluate<SUBST>(s, ls.wi);
    f3               contrib;
    if (mis) {
        const float light_pdf = ls.pdf * pick.pdf;
        const float weight = predividedPowerHeuristic(light_pdf, bx.pdf);
        contrib = S3(weight) * radiance * bx.reflection;
    } else {
        const float weight = 1.f / (pick.pdf * ls.pdf);
        contrib = S3(weight) * (S3(1.f) * radiance * bx.reflection);
    }
    P.pool.sh_o[slot] = f4(sorigin, slen);
    P.pool.sh_d[slot] = f4(sdir);
    P.pool.contrib[slot] = f4(contrib);
    return true;
}

// the tail of evaluateLight for the pick started in the previous round
ZD void finishLightPick(const Params& P, uint32_t slot, PathHead& h) {
    if (0 != P.pool.vis[slot]) {
        (void)h.smp.sample1D();  // light.zig:127 rs.stochastic_r: drawn only for unoccluded samples
        const float4 acc = P.pool.nee[slot], c = P.pool.contrib[slot];
        P.pool.nee[slot] = f4(xyz4(acc) + xyz4(c));
    }
    h.nee &= ~kNeePending;
}

struct NoTimer {
    ZD void mark(int) {}
};

// SUBST: the scene holds Substitute materials (zygpu_upload_scene picks the instantiation)
template <bool FIRST, bool SUBST>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_NEE_WAVES) wf_nee(Params P, uint32_t round) {
    const DScene&  s = c_scene;
    if (blockIdx.x * blockDim.x >= aliveBound(P, P.parity)) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at;
    bool           has_entry;
    if (!aliveOpen(P, P.parity, i, at, has_entry)) return;  // whole blocks take part in queuePush
    const bool     mis = ZYGPU_INTEGRATOR_PTMIS == c_view.v.integrator;
    bool     push = false;
    uint32_t slot = 0;
    if (has_entry) {
        slot = P.alive_in[at];
        if (FIRST) {
            PathHead h;
            loadHead(P, slot, h);
            const Surface sf = loadSurface(P, slot, h.flags);
            if (sf.ms.can_evaluate) {  // pathtracer_mis.zig:185-187: otherwise no draw at all
                const float     nee_threshold = splitThreshold(h.depth_surface);  // vertex.light_split_threshold of this vertex
                const float     select = h.smp.sample1D();
                const LightTree lt{s, s.light_tree};
                const uint32_t  stride = batchPaths();
                LightPick       first{0, 0.f};
                NoTimer         tm;
                const uint32_t  count = lt.randomLight(tm, sf.p, sf.ms.geo_n, sf.ms.translucent, select, nee_threshold,
                                                       [&](uint32_t k, LightPick pick) {
                                                           if (0 == k) first = pick;
                                                           else P.pool.picks[size_t(k) * stride + slot] = make_uint2(pick.offset, __float_as_uint(pick.pdf));
                                                       });
                h.nee = count & kNeeCountMask;
                if (count > 0) {
                    P.pool.nee[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (startLightPick<SUBST>(P, slot, h, sf, first, mis)) {
                        h.nee |= kNeePending;
                        push = true;
                    }
                }
                storeHead(P, slot, h);
            }
        } else {
            const uint32_t st = P.pool.q0.get(slot).w;
            if ((st & kNeePending) || (st & kNeeCountMask) > round) {
                PathHead h;
                loadHead(P, slot, h);
                if (h.nee & kNeePending) finishLightPick(P, slot, h);
                if ((h.nee & kNeeCountMask) > round) {
                    const Surface sf = loadSurface(P, slot, h.flags);
                    const uint2   pk = P.pool.picks[size_t(round) * batchPaths() + slot];
                    if (startLightPick<SUBST>(P, slot, h, sf, LightPick{pk.x, __uint_as_float(pk.y)}, mis)) {
                        h.nee |= kNeePending;
                        push = true;
                    }
                }
                storeHead(P, slot, h);
            }
        }
    }
    queuePush(P.q.shadow, &P.ctl->shadow_count[round & 1u], P.queue_capacity, push, slot);
}

// Scene.visibility (prop_tree.zig:185-240) for the shadow queue of one round
template <bool STATS>
__global__ void __launch_bounds__(256) wf_shadow(Params P, uint32_t round) {
    if (blockIdx.x * blockDim.x >= P.ctl->bound[P.parity]) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at;
    const uint32_t n = queueOpen(&P.ctl->shadow_count[round & 1u], P.queue_capacity, i, at);
    if (0 == blockIdx.x && threadIdx.x < 64) queueReset(&P.ctl->shadow_count[(round + 1u) & 1u]);  // the next round's wf_nee fills it
    Counters cnt{0, 0, 0};
    if (i < n) {
        const uint32_t slot = P.q.shadow[at];
        const float4   o = P.pool.sh_o[slot], d = P.pool.sh_d[slot];
        Ray            shadow = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
        P.pool.vis[slot] = sceneVisibility<STATS, true>(c_scene, shadow, cnt) ? 1u : 0u;
    }
    if (STATS) flushCounters(P, 2, cnt, (0 == blockIdx.x && 0 == threadIdx.x) ? n : 0);
}

// The large meshes' any-hit walks for the shadow rays wf_scatter_h suspended (scenes of the prop tables with a large mesh,
// relaxed units): a persistent kernel whose lanes fetch their rays from the round's queue themselves.  A shadow ray that grazes a
// million-triangle mesh walks hundreds of nodes, one that is blocked right away a few: with one ray per lane a wave waits for
// its longest walk (measured on configs[2]: wf_scatter spent 85 % of its wave-cycles in the walk or waiting for it at the wave's
// reconvergence and at the block's queue append).  Here a lane whose ray is decided takes the next one (the wave refills once
// ZYG_SHADOW_U_REFILL lanes are idle, one atomic per refill), and nothing but the walk is in the kernel's registers.
#ifndef ZYG_SHADOW_U_WAVES
#define ZYG_SHADOW_U_WAVES 6
#endif
#ifndef ZYG_SHADOW_U_REFILL
#define ZYG_SHADOW_U_REFILL 16
#endif
#ifndef ZYG_SHADOW_U_STEPS
#define ZYG_SHADOW_U_STEPS 8
#endif
template <bool WIDE>  // (a template so that only the unit that instantiates it emits it)
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_SHADOW_U_WAVES) wf_shadow_u(Params P, uint32_t round) {
    const DScene&  s = c_scene;
    const uint32_t  n = P.ctl->shadow_count[round & 1u].n[0];  // wf_scatter_h appends at the front only
    const uint32_t* queue = (round & 1u) ? P.q.heavy : P.q.shadow;
    if (0 == blockIdx.x && threadIdx.x < 64) queueReset(&P.ctl->shadow_count[(round + 1u) & 1u]);  // the next round's wf_scatter_h fills it
    const uint32_t lane = __lane_id();
    bool           have = false, exhausted = false;
    uint32_t       slot = 0, entry = 0;
    Ray            local = Ray::init(S3(0.f), S3(0.f), 0.f, 0.f), world = local;
    WideSpill<false> spill;
    AnyWalk4q      walk(spill);
    // the next large mesh at or behind solid_table[entry] whose bounds the ray enters: start its walk, or the ray is unoccluded
    auto nextMesh = [&]() {
        for (; entry < s.num_solid_table; ++entry) {
            const TlasEntry& e = s.solid_table[entry];
            if (0 == (e.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW) || !aabbIntersect(e.bmin, e.bmax, world)) continue;
            local = worldToObjectRay(entryTrafo(e), world);
            walk.start(s.meshes[e.mesh]);
            return true;
        }
        return false;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const uint32_t           num_idle = uint32_t(__popcll(idle));
        if (!exhausted && num_idle >= uint32_t(ZYG_SHADOW_U_REFILL)) {
            const int first = __ffsll(static_cast<long long>(idle)) - 1;
            uint32_t  base = 0;
            if (int(lane) == first) base = atomicAdd(&P.ctl->shadow_fetch, num_idle);
            base = __shfl(base, first);
            exhausted = base + num_idle >= n;
            if (!have) {
                const uint32_t i = base + uint32_t(__popcll(idle & ((1ull << lane) - 1ull)));
                if (i < n) {
                    slot = queue[i];
                    const float4 o = P.pool.sh_o[slot], d = P.pool.sh_d[slot];
                    world = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
                    entry = s.first_large;
                    have = nextMesh();
                    if (!have) P.pool.vis[slot] = 1u;
                }
            }
        }
        if (0 == __ballot(have)) {
            if (exhausted) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < ZYG_SHADOW_U_STEPS; ++it) {
            if (have) {
                const int r = walk.step(local);
                if (kWalkHit == r) {
                    P.pool.vis[slot] = 0u;
                    have = false;
                } else if (kWalkDone == r) {
                    entry += 1;
                    have = nextMesh();
                    if (!have) P.pool.vis[slot] = 1u;
                }
            }
        }
    }
}

// The closest hit of the extension rays that enter a large mesh's bounds (the back entries of the tick's extension queue), in
// the same form: a persistent kernel, a lane takes the next ray when its own is done.  The rest of the scene (Rectangles, small
// meshes) is tested when the ray is fetched and shortens it for the walks; the complete hit goes to the path's hit block, and the
// tick's wf_trace_shade (Params::mesh_pass) shades these entries from there instead of tracing them.
#ifndef ZYG_MESH_CLOSEST_WAVES
#define ZYG_MESH_CLOSEST_WAVES 5
#endif
template <bool WIDE>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_MESH_CLOSEST_WAVES) wf_mesh_closest(Params P, uint32_t) {
    const DScene&  s = c_scene;
    const uint32_t cur = P.parity;
    const uint32_t n = P.ctl->ext_count[cur].n[kCountStride];  // the queue's back end
    const uint32_t lane = __lane_id();
    bool           have = false, exhausted = false;
    uint32_t       slot = 0, entry = 0, depth = 0;
    Ray            local = Ray::init(S3(0.f), S3(0.f), 0.f, 0.f), world = local;
    Hit            hit;
    WideSpill<true> spill;
    ClosestWalk4q  walk(spill);
    auto nextMesh = [&]() {
        for (; entry < s.num_solid_table; ++entry) {
            const TlasEntry& e = s.solid_table[entry];
            if (!propVisibleAtDepth(e.flags, depth) || !aabbIntersect(e.bmin, e.bmax, world)) continue;
            local = worldToObjectRay(entryTrafo(e), world);
            walk.start(s.meshes[e.mesh]);
            return true;
        }
        return false;
    };
    auto finish = [&]() {
        if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
        P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
        P.pool.hit_prop[slot] = hit.prop;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const uint32_t           num_idle = uint32_t(__popcll(idle));
        if (!exhausted && num_idle >= uint32_t(ZYG_SHADOW_U_REFILL)) {
            const int first = __ffsll(static_cast<long long>(idle)) - 1;
            uint32_t  base = 0;
            if (int(lane) == first) base = atomicAdd(&P.ctl->mesh_fetch, num_idle);
            base = __shfl(base, first);
            exhausted = base + num_idle >= n;
            if (!have) {
                const uint32_t i = base + uint32_t(__popcll(idle & ((1ull << lane) - 1ull)));
                if (i < n) {
                    slot = P.q.ext[cur][P.queue_capacity - 1u - i];
                    const float4 o = P.pool.ray_o[slot], d = P.pool.ray_d[slot];
                    depth = (P.pool.q0.get(slot).y >> 3) & 31u;
                    world = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
                    hit.prop = kNull;
                    tableClosestBeforeLargeMeshes(s, world, depth, hit);
                    entry = s.first_large;
                    have = nextMesh();
                    if (!have) finish();
                }
            }
        }
        if (0 == __ballot(have)) {
            if (exhausted) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < ZYG_SHADOW_U_STEPS; ++it) {
            if (have) {
                const uint32_t prop = s.solid_table[entry].prop;
                const int      r = walk.step(local, [&](uint32_t triangle, float t, float u, float v) {
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    hit.primitive = triangle;
                    hit.prop = prop;
                    world.max_t = t;
                });
                if (kWalkDone == r) {
                    entry += 1;
                    have = nextMesh();
                    if (!have) finish();
                }
            }
        }
    }
}

// ---- The same two persistent walkers over the eight-wide compressed tree (bvh8.hpp, dscene.hpp Walk8; option "bvh8").  What
// changes against the four-wide form above: a step is a node step for every lane that has a node to look at (eight children per 80 B
// node, hit bits in front-to-back order by the ray's octant: no sorting network, one stack entry per node), then ONE triangle for the
// lanes that have triangles pending, but only once ZYG_U8_POSTPONE lanes of the wave have some (or no lane has a node left): until
// then a lane parks its triangle group on its stack and goes on with nodes, so that the triangle test runs with many lanes instead
// of serialising the wave behind whichever lane happens to stand at a leaf.
#ifndef ZYG_U8_POSTPONE
#define ZYG_U8_POSTPONE 4  // of 64 lanes
#endif
#ifndef ZYG_U8_EAGER
#define ZYG_U8_EAGER 0  // 1: a lane with a node to look at takes its node step every iteration, parking what is left of its triangle group
#endif
#ifndef ZYG_U8_STEPS
#define ZYG_U8_STEPS 16
#endif
#ifndef ZYG_U8_REFILL
#define ZYG_U8_REFILL 32
#endif
#ifndef ZYG_SHADOW_U8_WAVES
#define ZYG_SHADOW_U8_WAVES 4
#endif
#ifndef ZYG_MESH_CLOSEST8_WAVES
#define ZYG_MESH_CLOSEST8_WAVES 4
#endif
template <bool WIDE>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_SHADOW_U8_WAVES) wf_shadow_u8(Params P, uint32_t round) {
    const DScene&   s = c_scene;
    const uint32_t  n = P.ctl->shadow_count[round & 1u].n[0];
    const uint32_t* queue = (round & 1u) ? P.q.heavy : P.q.shadow;
    if (0 == blockIdx.x && threadIdx.x < 64) queueReset(&P.ctl->shadow_count[(round + 1u) & 1u]);
    const uint32_t lane = __lane_id();
    bool           have = false, exhausted = false;
    uint32_t       slot = 0, entry = 0;
    Ray            local = Ray::init(S3(0.f), S3(0.f), 0.f, 0.f), world = local;
    GroupSpill     spill;
    Walk8          walk(spill);
    auto nextMesh = [&]() {
        for (; entry < s.num_solid_table; ++entry) {
            const TlasEntry& e = s.solid_table[entry];
            if (0 == (e.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW) || !aabbIntersect(e.bmin, e.bmax, world)) continue;
            local = worldToObjectRay(entryTrafo(e), world);
            walk.start(s.meshes[e.mesh], local);
            return true;
        }
        return false;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const uint32_t           num_idle = uint32_t(__popcll(idle));
        if (!exhausted && num_idle >= uint32_t(ZYG_U8_REFILL)) {
            const int first = __ffsll(static_cast<long long>(idle)) - 1;
            uint32_t  base = 0;
            if (int(lane) == first) base = atomicAdd(&P.ctl->shadow_fetch, num_idle);
            base = __shfl(base, first);
            exhausted = base + num_idle >= n;
            if (!have) {
                const uint32_t i = base + uint32_t(__popcll(idle & ((1ull << lane) - 1ull)));
                if (i < n) {
                    slot = queue[i];
                    const float4 o = P.pool.sh_o[slot], d = P.pool.sh_d[slot];
                    world = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
                    entry = s.first_large;
                    have = nextMesh();
                    if (!have) P.pool.vis[slot] = 1u;
                }
            }
        }
        if (0 == __ballot(have)) {
            if (exhausted) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < ZYG_U8_STEPS; ++it) {
            if (have && walk.nodePending()) {  // (a node step makes a new triangle group: an unfinished one waits on the stack meanwhile)
                if (ZYG_U8_EAGER && walk.trianglesPending() && walk.stack.sp < kStack8Lds - 4) {
                    walk.stack.push(walk.tg);
                    walk.tg.y = 0u;
                }
                if (!walk.trianglesPending()) walk.nodeStep(local.min_t, local.max_t);
            }
            const bool               tri = have && walk.trianglesPending();
            const unsigned long long tri_lanes = __ballot(tri);
            if (tri_lanes) {
                // (a lane whose stack is filling up does not park another group: the node groups below need the room)
                const bool test = __popcll(tri_lanes) >= ZYG_U8_POSTPONE || 0 == __ballot(have && walk.nodePending());
                if (tri) {
                    if (test || walk.stack.sp >= kStack8Lds - 4) {
                        f3       a, b, c;
                        uint32_t reference;
                        float    t, u, v;
                        walk.nextTriangle(a, b, c, reference);
                        if (triangleIntersect(local, a, b, c, t, u, v)) {
                            P.pool.vis[slot] = 0u;
                            have = false;
                        }
                    } else {
                        walk.stack.push(walk.tg);
                        walk.tg.y = 0u;
                    }
                }
            }
            if (have && !walk.nodePending() && !walk.trianglesPending()) {
                if (walk.stack.sp > 0) {
                    walk.adopt(walk.stack.pop());
                } else {
                    entry += 1;
                    have = nextMesh();
                    if (!have) P.pool.vis[slot] = 1u;
                }
            }
        }
    }
}

template <bool WIDE>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_MESH_CLOSEST8_WAVES) wf_mesh_closest8(Params P, uint32_t) {
    const DScene&  s = c_scene;
    const uint32_t cur = P.parity;
    const uint32_t n = P.ctl->ext_count[cur].n[kCountStride];  // the queue's back end
    const uint32_t lane = __lane_id();
    bool           have = false, exhausted = false;
    uint32_t       slot = 0, entry = 0, depth = 0;
    Ray            local = Ray::init(S3(0.f), S3(0.f), 0.f, 0.f), world = local;
    Hit            hit;
    GroupSpill     spill;
    Walk8          walk(spill);
    auto nextMesh = [&]() {
        for (; entry < s.num_solid_table; ++entry) {
            const TlasEntry& e = s.solid_table[entry];
            if (!propVisibleAtDepth(e.flags, depth) || !aabbIntersect(e.bmin, e.bmax, world)) continue;
            local = worldToObjectRay(entryTrafo(e), world);
            walk.start(s.meshes[e.mesh], local);
            return true;
        }
        return false;
    };
    auto finish = [&]() {
        if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
        P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
        P.pool.hit_prop[slot] = hit.prop;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const uint32_t           num_idle = uint32_t(__popcll(idle));
        if (!exhausted && num_idle >= uint32_t(ZYG_U8_REFILL)) {
            const int first = __ffsll(static_cast<long long>(idle)) - 1;
            uint32_t  base = 0;
            if (int(lane) == first) base = atomicAdd(&P.ctl->mesh_fetch, num_idle);
            base = __shfl(base, first);
            exhausted = base + num_idle >= n;
            if (!have) {
                const uint32_t i = base + uint32_t(__popcll(idle & ((1ull << lane) - 1ull)));
                if (i < n) {
                    slot = P.q.ext[cur][P.queue_capacity - 1u - i];
                    const float4 o = P.pool.ray_o[slot], d = P.pool.ray_d[slot];
                    depth = (P.pool.q0.get(slot).y >> 3) & 31u;
                    world = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
                    hit.prop = kNull;
                    // (the Rectangles and small meshes in front of the large ones: wf_trace_shade, Params::mesh_pass 2)
                    entry = s.first_large;
                    have = nextMesh();
                    if (!have) finish();
                }
            }
        }
        if (0 == __ballot(have)) {
            if (exhausted) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < ZYG_U8_STEPS; ++it) {
            if (have && walk.nodePending()) {  // (a node step makes a new triangle group: an unfinished one waits on the stack meanwhile)
                if (ZYG_U8_EAGER && walk.trianglesPending() && walk.stack.sp < kStack8Lds - 4) {
                    walk.stack.push(walk.tg);
                    walk.tg.y = 0u;
                }
                if (!walk.trianglesPending()) walk.nodeStep(local.min_t, local.max_t);
            }
            const bool               tri = have && walk.trianglesPending();
            const unsigned long long tri_lanes = __ballot(tri);
            if (tri_lanes) {
                // (a lane whose stack is filling up does not park another group: the node groups below need the room)
                const bool test = __popcll(tri_lanes) >= ZYG_U8_POSTPONE || 0 == __ballot(have && walk.nodePending());
                if (tri) {
                    if (test || walk.stack.sp >= kStack8Lds - 4) {
                        f3       a, b, c;
                        uint32_t reference;
                        float    t, u, v;
                        walk.nextTriangle(a, b, c, reference);
                        // (the order a lane meets its triangles in hangs on what the wave's other lanes do: of two triangles at one
                        // distance, a ray through their shared edge, the one the reference's order names first wins, whatever the order)
                        const uint32_t prop = s.solid_table[entry].prop;
                        if (triangleIntersect(local, a, b, c, t, u, v) && (t < local.max_t || prop != hit.prop || reference < hit.primitive)) {
                            local.max_t = t;
                            world.max_t = t;
                            hit.t = t;
                            hit.u = u;
                            hit.v = v;
                            hit.primitive = reference;
                            hit.prop = prop;
                        }
                    } else {
                        walk.stack.push(walk.tg);
                        walk.tg.y = 0u;
                    }
                }
            }
            if (have && !walk.nodePending() && !walk.trianglesPending()) {
                if (walk.stack.sp > 0) {
                    walk.adopt(walk.stack.pop());
                } else {
                    entry += 1;
                    have = nextMesh();
                    if (!have) finish();
                }
            }
        }
    }
}

// The bxdf sample and the path extension of a vertex whose light samples are done (pathtracer_mis.zig:123-166): the tail of
// wf_extend and of wf_scatter_h.  `push`: the path goes on (its slot belongs into the next tick's extension queue, at the back
// if the new ray enters a large mesh's bounds).
template <bool SUBST, bool EXTRA>
ZD void extendVertex(const Params& P, const DScene& s, uint32_t slot, PathHead& h, const Surface& sf, f3 throughput, bool& push, bool& back) {
    const MaterialSample& ms = sf.ms;
    const bool            mis = ZYGPU_INTEGRATOR_PTMIS == c_view.v.integrator;
    BxdfSample            bs;
    if (!ms.template sample<SUBST, EXTRA>(s, h.smp, bs)) {
        finalize(P, slot, S3(0.f), h.flags, h.smp.rng);
        return;
    }
    // Vertex.State.update (vertex.zig:30-43); every sample in scope is a reflection (never .Straight)
    if (kScatterSpecular == bs.scattering) {
        h.flags |= ST_SPECULAR;
        h.flags = (0.f == bs.reg_alpha) ? (h.flags | ST_SINGULAR) : (h.flags & ~uint32_t(ST_SINGULAR));
        if (h.flags & ST_PRIMARY) h.flags |= ST_STARTED_SPECULAR;
    } else {
        h.flags &= ~uint32_t(ST_SPECULAR | ST_SINGULAR | ST_PRIMARY);
    }
    const bool packed_prev = 0 != P.packed_prev;
    float      bxdf_pdf = P.pool.ray_d.get(slot).w;
    if (mis) {
        h.flags = ms.translucent ? (h.flags | ST_TRANSLUCENT) : (h.flags & ~uint32_t(ST_TRANSLUCENT));
        h.vdepth = h.depth_surface;
        bxdf_pdf = bs.pdf;
        if (!packed_prev) P.pool.vgeo_n[slot] = f4(ms.geo_n, bs.reg_alpha);
    }
    throughput = throughput * (bs.reflection / S3(bs.pdf));
    const f3 new_origin = surfaceOffsetP(sf.p, sf.frag_geo_n, bs.wi);
    h.depth_surface += 1;
    h.flags &= ~uint32_t(ST_TRANSPARENT);  // transparent && (Transmission || Straight): a reflection clears it
    h.smp.incrementPadding();

    P.pool.ray_o[slot] = f4(new_origin, kRayMaxT);
    P.pool.ray_d[slot] = f4(bs.wi, bxdf_pdf);
    back = entersMeshBounds(new_origin, bs.wi);
    if (packed_prev) {  // (only PathtracerMIS looks at geo_n and the roughness: the other integrators leave them behind)
        const PrevVertex prev{sf.p, mis ? ms.geo_n : S3(0.f), mis ? bs.reg_alpha : 0.f};
        P.pool.thr[slot] = f4(throughput, prev.alpha);
        storeHead(P, slot, h, &prev);
        storePrevTail(P, slot, prev);
    } else {
        P.pool.thr[slot] = f4(throughput);
        P.pool.vorigin[slot] = f4(sf.p);
        storeHead(P, slot, h);
    }
    push = true;
}

// ---------------------------------------------------------------- wf_extend
// the last pick's tail, the NEE sum into the IValue, then the bxdf sample and the path extension (pathtracer_mis.zig:123-166)
// EXTRA: the scene runs the MESHL instantiations (PathTracer has no light sampling, so it ends its ticks here whatever the scene)
template <bool SUBST, bool EXTRA>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_EXTEND_WAVES) wf_extend(Params P) {
    const DScene&  s = c_scene;
    const uint32_t nxt = P.parity ^ 1u;
    if (blockIdx.x * blockDim.x >= aliveBound(P, P.parity)) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at;
    bool           has_entry;
    if (!aliveOpen(P, P.parity, i, at, has_entry)) return;
    bool     push = false, back = false;
    uint32_t slot = 0;
    if (has_entry) {
        slot = P.alive_in[at];
        PathHead h;
        loadHead(P, slot, h);
        const Surface sf = loadSurface<false, EXTRA>(P, slot, h.flags);
        f3            throughput = xyz4(P.pool.thr[slot]);
        if (EXTRA) skipTextureDraws(P, slot, sf.material, h);
        if (EXTRA && c_view.v.aov_slots) commonAov(P, slot, slot, h, sf, throughput);

        if (h.nee & kNeePending) finishLightPick(P, slot, h);
        if (h.nee & kNeeCountMask) ivalueAdd<true>(P, slot, throughput * xyz4(P.pool.nee[slot]), h.depth_surface, 1u, false, false);
        h.nee = 0;

        extendVertex<SUBST, EXTRA>(P, s, slot, h, sf, throughput, push, back);
    }
    queuePush(P.q.ext[nxt], &P.ctl->ext_count[nxt], P.queue_capacity, push, slot, back);
}

// ---------------------------------------------------------------- wf_scatter_h
// wf_scatter for the scenes of the prop tables with a large mesh (relaxed units; option "mesh_kernels"): the light samples of a
// vertex are taken in place as wf_scatter takes them, but a shadow ray is traced in place only through the cheap part of the
// scene (the table's Rectangles and small meshes).  If none of them blocks it and it enters a large mesh's bounds, the vertex
// is suspended: the ray, the pick's pre-weighted contribution (neither depends on the visibility), the sampler and the sum so
// far go to the pool, the slot into the round's suspended queue, and wf_shadow_u walks the meshes for all suspended rays with its
// lanes kept busy.  The next launch (RESUME) takes up the suspended vertices only: it finishes the pending pick with the answer
// (the draw that follows an unoccluded sample, light.zig:127), goes on with the vertex' remaining picks (which may suspend it
// again) and then, like wf_scatter, samples the bxdf and extends the path.  A vertex whose shadow rays never enter a large mesh's
// bounds is done in the first launch: configs[2] with the mesh on the floor (few such rays) keeps wf_scatter's speed, with the
// mesh in the middle of the box the divergent walks leave the kernel whose lanes would wait for them (measured: DESIGN 4).
// Rectangle lights with one sample per pick (as the wf_nee rounds).  The suspended queues ping-pong between q.shadow and
// q.heavy (wf_shade's second pass is done with it by now); their lengths are Control::shadow_count[round & 1].
constexpr uint32_t kNeeNextShift = 8u;  // PathHead::nee [8:14]: the pick a suspended vertex goes on with

template <bool SUBST>
ZD bool entersLargeMeshBounds(const DScene& s, const Ray& ray) {
    bool enters = false;
    for (uint32_t k = s.first_large; k < s.num_solid_table; ++k) {
        const TlasEntry& e = s.solid_table[k];
        enters = enters || (0 != (e.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW) && aabbIntersect(e.bmin, e.bmax, ray));
    }
    return enters;
}

template <bool RESUME, bool SUBST>
__global__ void __launch_bounds__(256) ZYG_WAVES(SUBST ? ZYG_SCATTER_BIG_WAVES : ZYG_SCATTER_WAVES) wf_scatter_h(Params P, uint32_t round) {
    const DScene&  s = c_scene;
    const uint32_t nxt = P.parity ^ 1u;
    if (blockIdx.x * blockDim.x >= (RESUME ? P.ctl->bound[P.parity] : aliveBound(P, P.parity))) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at = i;
    // RESUME: the vertices the previous round suspended (its queue's front), else the tick's alive queue
    const uint32_t* in = RESUME ? (((round + 1u) & 1u) ? P.q.heavy : P.q.shadow) : P.alive_in;
    uint32_t*       suspended_queue = (round & 1u) ? P.q.heavy : P.q.shadow;
    const bool      mis = ZYGPU_INTEGRATOR_PTMIS == c_view.v.integrator;
    if (0 == i) P.ctl->shadow_fetch = 0;  // the round's wf_shadow_u hands out its queue from the start
    bool has_entry;
    if (RESUME) {
        const uint32_t total = P.ctl->shadow_count[(round + 1u) & 1u].n[0];
        if (i >= ((total + 255u) & ~255u)) return;  // whole blocks take part in the queue appends
        has_entry = i < total;
    } else {
        // Params::alive_key_mask: the sub-queues (= hit props, modulo kAliveKeys) this launch takes; a scene with Substitute and other
        // materials runs the lighter instantiation on the sub-queues that hold no Substitute prop (a block reads one sub-queue)
        uint32_t key = 0;
        if (!aliveOpen(P, P.parity, i, at, has_entry, &key) || 0 == ((P.alive_key_mask >> key) & 1u)) return;
    }
    bool     push = false, back = false, suspend = false;
    uint32_t slot = 0;
    if (has_entry) {
        slot = in[at];
        PathHead h;
        loadHead(P, slot, h);
        const Surface         sf = loadSurface(P, slot, h.flags);
        const MaterialSample& ms = sf.ms;
        const f3              throughput = xyz4(P.pool.thr[slot]);
        const uint32_t        stride = batchPaths();
        f3                    nee = S3(0.f);
        uint32_t              count = 0, pk = 0;
        LightPick             first{0, 0.f};
        if (RESUME) {
            count = h.nee & kNeeCountMask;
            pk = (h.nee >> kNeeNextShift) & kNeeCountMask;
            nee = xyz4(P.pool.nee[slot]);
            if (0 != P.pool.vis[slot]) {
                (void)h.smp.sample1D();  // light.zig:127 rs.stochastic_r: drawn only for unoccluded samples
                nee = nee + xyz4(P.pool.contrib[slot]);
            }
        } else if (ZYGPU_INTEGRATOR_PT != c_view.v.integrator && ms.can_evaluate) {  // pathtracer_mis.zig:185-187: otherwise no draw at all
            const float     nee_threshold = splitThreshold(h.depth_surface);
            const float     select = h.smp.sample1D();
            const LightTree lt{s, s.light_tree};
            NoTimer         tm;
            count = lt.randomLight(tm, sf.p, ms.geo_n, ms.translucent, select, nee_threshold, [&](uint32_t k, LightPick pick) {
                if (0 == k) first = pick;
                else P.pool.picks[size_t(k) * stride + slot] = make_uint2(pick.offset, __float_as_uint(pick.pdf));
            });
        }
        for (; pk < count; ++pk) {
            LightPick pick = first;
            if (RESUME || pk > 0) {
                const uint2 stored = P.pool.picks[size_t(pk) * stride + slot];
                pick = LightPick{stored.x, __uint_as_float(stored.y)};
            }
            // evaluateLight up to the shadow test (startLightPick, with the ray and the contribution still in registers)
            const ZygpuLight     light = s.lights[pick.offset];
            const Trafo          trafo = loadTrafo(s, light.prop);
            const ZygpuMaterial& lm = s.materials[s.material_ids[s.props[light.prop].parts_start + light.part]];
            SampleTo             ls;
            if (!rectangleSampleTo(sf.p, ms.geo_n, trafo, 0 != light.two_sided, ms.translucent, h.smp, ls)) continue;
            const f3 sorigin = surfaceOffsetP(sf.p, sf.frag_geo_n, ls.wi);
            f3       sdir;
            float    slen;
            shadowRay(sorigin, ls, sdir, slen);
            const Ray shadow = Ray::init(sorigin, sdir, 0.f, slen);
            if (tableBlocksBeforeLargeMeshes(s, shadow)) continue;  // occluded: no contribution, no draw
            const f3         radiance = materialRadiance(s, lm, ls.wi, trafo, light.prop, light.part, false);
            const BxdfResult bx = ms.template evaluate<SUBST>(s, ls.wi);
            f3               contrib;
            if (mis) {
                const float light_pdf = ls.pdf * pick.pdf;
                const float weight = predividedPowerHeuristic(light_pdf, bx.pdf);
                contrib = S3(weight) * radiance * bx.reflection;
            } else {
                const float weight = 1.f / (pick.pdf * ls.pdf);
                contrib = S3(weight) * (S3(1.f) * radiance * bx.reflection);
            }
            if (!entersLargeMeshBounds<SUBST>(s, shadow)) {  // unoccluded
                (void)h.smp.sample1D();
                nee = nee + contrib;
                continue;
            }
            P.pool.sh_o[slot] = f4(sorigin, slen);
            P.pool.sh_d[slot] = f4(sdir);
            P.pool.contrib[slot] = f4(contrib);
            suspend = true;
            break;
        }
        if (suspend) {
            h.nee = (count & kNeeCountMask) | kNeePending | ((pk + 1u) << kNeeNextShift);
            P.pool.nee[slot] = f4(nee);
            storeHead(P, slot, h);
        } else {
            if (count > 0) ivalueAdd<true>(P, slot, throughput * nee, h.depth_surface, 1u, false, false);
            h.nee = 0;
            extendVertex<SUBST, false>(P, s, slot, h, sf, throughput, push, back);
        }
    }
    queuePush(suspended_queue, &P.ctl->shadow_count[round & 1u], P.queue_capacity, suspend, slot);
    __syncthreads();  // queuePush's shared words are reused
    queuePush(P.q.ext[nxt], &P.ctl->ext_count[nxt], P.queue_capacity, push, slot, back);
}

// ---------------------------------------------------------------- wf_scatter
// The in-place alternative to wf_nee / wf_shadow / wf_extend (option "nee_in_place"): sampleLights with the shadow rays
// traced inside the pick loop, then the bxdf sample and the path extension, in one kernel.  Fewer launches and no state
// round trips, but every lane waits for the longest pick loop of its wave and the kernel needs ~240 VGPRs.
// SUBST: the scene holds Substitute materials (zygpu_upload_scene picks the instantiation)
// MULTI: a light has emittance num_samples > 1; MESHL: the scene has TriangleMesh lights
// POOL: `slot` below is a vertex index, `ps` the camera sample it belongs to; Glass may return two bxdf samples
// MASKS: a material of the scene has an opacity mask (shadow rays evaluate it); always on in the STATS instantiations
// UTLAS: shadow rays walk the scene's prop table (dscene.hpp TlasEntry; not with STATS, MESHL, POOL or MASKS)
// what a vertex of wf_scatter hands to the queues
struct ScatterOut {
    bool     push = false, back = false;
    // POOL: the children of this vertex: their vertex index, their place in the next level (= the round that handles them)
    uint32_t child[2] = {0, 0}, child_round[2] = {kNull, kNull};
    bool     child_back[2] = {false, false};
};

// One vertex of wf_scatter: sampleLights / evaluateLight with the shadow rays traced in the pick loop, then the bxdf sample
// and the path extension (pathtracer_mis.zig:123-278).  Template parameters as wf_scatter's.
// PHASE (the POOL scenes' complete instantiation only; 0 everywhere else: the vertex in one piece):
//   1 (wf_light): the light half without a closure: light-tree picks, the light kinds' sampling, the shadow rays' walks, the lights'
//     radiance and every sampler draw of sampleLights, in the reference's order; what each light sample that reached its light needs for
//     its closure evaluation goes to Params::nee_rec, the sampler goes back to the pool.  The closure is built for the two bits that
//     steer the light loop (can_evaluate, translucent) and dropped: the walk runs with ~100 registers fewer.
//   2 (wf_scatter<.., 2>): the closure half: evaluates the stored samples in their order (same sums in the same grouping: the film is
//     the same bits), then the bxdf sample and the path extension.
constexpr uint32_t kMaxNeeStored = 4, kNeeOverflow = 0xFFFFFFFFu;
template <bool STATS, bool SUBST, bool MULTI, bool MESHL, bool POOL, bool MASKS, bool UTLAS, bool WIDE, int PHASE = 0, uint32_t KINDS = kAllLightKinds>
ZD void scatterVertex(const Params& P, const DScene& s, uint32_t slot, ScatterOut& out, Counters& cnt, unsigned long long& shadow_rays,
                      SectionTimer& tm, const HitRecord* record = nullptr, bool unlit = false) {
    const uint32_t nxt = P.parity ^ 1u;
    const uint32_t kind = c_view.v.integrator;
    const bool     mis = ZYGPU_INTEGRATOR_PTMIS == kind;
    bool&          push = out.push;
    bool&          back = out.back;
    uint32_t(&child)[2] = out.child;
    uint32_t(&child_round)[2] = out.child_round;
    bool(&child_back)[2] = out.child_back;
    (void)nxt;
    const uint32_t ps = sampleSlot<POOL>(slot);
    PathHead h;
    loadHead<POOL>(P, slot, h);
    // -DZYG_REBUILD_SURFACE=1 (an experiment, kept): the instantiations that know every light kind build the closure again where it
    // is used (for a light sample that reached its light; for the bxdf sample behind the light loop) instead of keeping its ~100
    // registers alive across the light kinds' sampling code and the shadow ray's walk, where the register allocator spills them
    // (wf_scatter<.., MESHL, POOL>: 554 spilled registers).  It trades spill traffic for the closure's set-up: no gain measured.
#ifndef ZYG_REBUILD_SURFACE
#define ZYG_REBUILD_SURFACE 0  // measured (gpurun_out/r04f, one call): cornell_glass 115.3 -> 113.6, the outdoor scene at 3840x2160 218.3 -> 224.8 Msamples/s: off
#endif
    constexpr bool        kRebuild = MESHL && 0 != ZYG_REBUILD_SURFACE;
    Surface               sf = loadSurface<POOL, MESHL>(P, slot, h.flags, record);
    const MaterialSample& ms = sf.ms;
    const float4          thr4 = P.pool.thr[slot];
    f3                    throughput = xyz4(thr4);
    const float           split_weight = POOL ? thr4.w : 1.f;
    // The sum this kernel may add to (IValue.add, helper.zig:11-19: the direct sum for the camera vertex' light samples, the indirect one
    // behind it), read here, with the path's line, instead of where the light's contribution is known: there the add waited for a load
    // of its own (a fifth of the kernel's wave cycles on configs[1], tools/sections.py)
    constexpr bool kPreloadSum = !POOL && 0 == PHASE && kPackedAccumulators;
    const bool     preload_sum = kPreloadSum && 0 != P.packed_acc;
    const bool     sum_is_direct = h.depth_surface < 1u;
    float4*        sum_at = nullptr;
    float4         sum_old = make_float4(0.f, 0.f, 0.f, 0.f);
    if (preload_sum) {
        sum_at = sum_is_direct ? reinterpret_cast<float4*>(pathBlock(P, slot) + 112) : recordAddress(P.pool.indirect, slot);
        sum_old = *sum_at;
    }
    if (!POOL && record && record->lazy) {
        // the first shading pass drew twice and wrote nothing back (shadeVertex `lazy_out`): russianRoulette (helper.zig:75-89: the
        // path is here, so it survived) and rs.stochastic_r (vertex.zig:165: its value came with the hit record), from the line as
        // the previous wf_scatter left it
        (void)h.smp.sample1D();
        const float continuation_probability = hmax3(throughput) / 0.1f;
        if (continuation_probability < 1.f) throughput = throughput / S3(continuation_probability);
        (void)h.smp.sample1D();
    }
    const uint32_t texture_draws = (MESHL && 2 != PHASE) ? skipTextureDraws(P, slot, sf.material, h) : 0u;

    // (`unlit`: the vertex came through the alive queue's back end and skipped wf_light: it has no stored samples, P.nee_count is another vertex')
    if (2 == PHASE && !unlit && kNeeOverflow == P.nee_count[ps]) return;  // wf_light sent the vertex through the unsplit kernel
    if (MESHL && 1 != PHASE && c_view.v.aov_slots) commonAov(P, slot, ps, h, sf, throughput, record);
    tm.mark(0);

    // POOL: the split budget of this vertex (Pool.maxSplits, vertex.zig:305-308; PathtracerMIS only: pathtracer.zig:73,
    // pathtracer_dl.zig:82, 160): rough Glass evaluates light samples with it, Glass splits its bxdf sample by it
    uint32_t max_splits = 1;
    if (POOL && mis) {
        const uint32_t m = kPoolVertices / __float_as_uint(P.pool.vorigin.get(slot).w);
        max_splits = m - ((h.flags & ST_PRIMARY) ? 0u : min(h.depth_surface, m - 1u));
    }

    // PathtracerMIS on a shadow catcher (pathtracer_mis.zig:56, 106-115, 248-276): an occluded light sample is evaluated with tr = 0
    // instead of dropped, and what the shadow-catcher lights send is kept apart, with and without the occluders
    const bool catcher_scene = POOL && MESHL && mis && 0 != s.has_catcher;
    const bool shadow_catcher = catcher_scene && 0 != (s.props[P.pool.hit_prop[slot]].flags & ZYGPU_PROP_SHADOW_CATCHER);
    f3         catcher_occluded = S3(0.f), catcher_unoccluded = S3(0.f);

    // ---- sampleLights / evaluateLight (pathtracer_mis.zig:174-278, pathtracer_dl.zig:100-167)
#ifndef ZYG_EXPERIMENT_SCATTER_PART
#define ZYG_EXPERIMENT_SCATTER_PART 0  // register-pressure experiments (wrong images): 1 = the light samples alone, 2 = the bxdf sample alone
#endif
    uint32_t stored = 0;       // PHASE 1: light samples written to Params::nee_rec
    bool     overflow = false;  //          more of them than the record holds
    if (2 == PHASE) {
        // the light samples wf_light stored, in its order; a pick's samples are summed by themselves first (evaluateLight's `result`),
        // then added to the vertex' sum or, on a shadow catcher, to what the catcher lights send
        if (ZYGPU_INTEGRATOR_PT != kind && ms.can_evaluate) {
            const uint32_t count = unlit ? 0u : uint32_t(P.nee_count[ps]);
            f3             nee = S3(0.f), occluded = S3(0.f), unoccluded = S3(0.f);
            uint32_t       open_pick = kNull;
            bool           open_catcher_light = false;
            auto closePick = [&]() {
                if (kNull == open_pick) return;
                if (shadow_catcher && open_catcher_light) {  // pathtracer_mis.zig:274-276
                    catcher_occluded = catcher_occluded + occluded;
                    catcher_unoccluded = catcher_unoccluded + unoccluded;
                } else {
                    nee = nee + occluded;
                }
            };
            for (uint32_t k = 0; k < count; ++k) {
                const size_t at = size_t(3u * k) * batchPaths() + ps;
                const float4 a = P.nee_rec[at], b = P.nee_rec[at + batchPaths()], c = P.nee_rec[at + 2 * size_t(batchPaths())];
                if (__float_as_uint(b.w) != open_pick) {
                    closePick();
                    open_pick = __float_as_uint(b.w);
                    open_catcher_light = 0.f != c.w;
                    occluded = unoccluded = S3(0.f);
                }
                const f3         radiance = xyz4(b), tr = xyz4(c);
                const BxdfResult bx = ms.template evaluate<SUBST, POOL, MESHL>(s, xyz4(a), POOL && max_splits > 1);
                if (mis) {
                    const float weight = predividedPowerHeuristic(a.w, bx.pdf);
                    const f3    unocc = S3(weight) * radiance * bx.reflection;
                    if (catcher_scene) unoccluded = unoccluded + unocc;
                    occluded = occluded + tr * unocc;  // pathtracer_mis.zig:267-270
                } else {
                    const float weight = 1.f / a.w;
                    occluded = occluded + S3(weight) * (tr * radiance * bx.reflection);
                }
            }
            closePick();
            ivalueAdd<!POOL>(P, ps, (POOL ? throughput * S3(split_weight) : throughput) * nee, h.depth_surface, 1u, false, false);
        }
    } else
    if (ZYGPU_INTEGRATOR_PT != kind && ms.can_evaluate && 2 != ZYG_EXPERIMENT_SCATTER_PART) {  // pathtracer_mis.zig:185-187: otherwise no draw at all
        const float     nee_threshold = splitThreshold(h.depth_surface);  // vertex.light_split_threshold of this vertex
        const float     select = h.smp.sample1D();
        const LightTree lt{s, s.light_tree};
        NoTimer         notimer;
        f3              nee = S3(0.f);
        // the picks first (Scene.randomLightSpatial fills a buffer too, scene.zig:592-610), then one loop over them: the
        // lanes of a wave leave the tree walk together instead of interleaving it with their shadow rays
        // (the first pick, the only one unless the tree splits, in registers: the buffer is scratch)
        // (as two scalars: copies between LightPick objects are selected by address again, dmath.hpp pick)
        LightPick      picks[kMaxLightPicks];
        uint32_t       first_offset = 0;
        float          first_pdf = 0.f;
        const uint32_t num_picks = lt.randomLight(notimer, sf.p, ms.geo_n, ms.translucent, select, nee_threshold, [&](uint32_t k, LightPick pick) {
            if (0 == k) first_offset = pick.offset, first_pdf = pick.pdf;
            else picks[k] = pick;
        });
        tm.mark(1);
        for (uint32_t pk = 0; pk < num_picks; ++pk) {
            uint32_t pick_offset = first_offset;
            float    pick_pdf = first_pdf;
            if (pk > 0) pick_offset = picks[pk].offset, pick_pdf = picks[pk].pdf;
            const LightPick pick{pick_offset, pick_pdf};
            const ZygpuLight     light = s.lights[pick.offset];
            // the light at the path's time (light.zig:84-107 <- scene.propTransformationAt): an animated light moves
            const Trafo          trafo = propTrafoAt<MESHL>(s, light.prop, pathTime<POOL, MESHL>(P, slot));
            const ZygpuMaterial& lm = s.materials[s.material_ids[s.props[light.prop].parts_start + light.part]];
            // PathtracerDL samples its lights with threshold 0 (pathtracer_dl.zig:139-167)
            const float sample_threshold = mis ? nee_threshold : 0.f;
            f3          occluded = S3(0.f), unoccluded = S3(0.f);

            // evaluateLight's loop body for one light sample (pathtracer_mis.zig:248-272 / pathtracer_dl.zig:150-166)
            const uint32_t light_shape = MESHL ? s.props[light.prop].shape : uint32_t(ZYGPU_SHAPE_RECTANGLE);
            const bool     distant =
                ZYGPU_SHAPE_DISTANT == light_shape || ZYGPU_SHAPE_DOME == light_shape || ZYGPU_SHAPE_CANOPY == light_shape;
            auto connect = [&](const SampleTo& ls) {
                const f3 sorigin = surfaceOffsetP(sf.p, sf.frag_geo_n, ls.wi);
                f3       sdir;
                float    slen;
                shadowRay(sorigin, ls, sdir, slen, distant);
                Ray shadow = Ray::init(sorigin, sdir, 0.f, slen);
                shadow.time = pathTime<POOL, MESHL>(P, slot);
                tm.mark(2);
                if (STATS) shadow_rays += 1;
                f3 tr = S3(1.f);  // POOL: thin Glass lets shadow rays through, attenuated (Scene.visibility's tr)
                // MESHL: the scene may have opacity masks, whose draws come from the path's sampler (material.zig:238-245)
                auto draw = [&]() { return h.smp.sample1D(); };
                constexpr bool kMasks = MASKS && MESHL;
#ifndef ZYG_EXPERIMENT_NEE_SKIP
#define ZYG_EXPERIMENT_NEE_SKIP 0  // register-pressure experiments (wrong images): 1 no shadow test, 2 no closure evaluation, 4 Rectangle lights only, 8 no radiance
#endif
                const bool visible = (ZYG_EXPERIMENT_NEE_SKIP & 1) ? true
                                     : UTLAS              ? sceneVisibilityUniform<WIDE, 0 != ZYG_FAST>(s, shadow)
                                     : (POOL || kMasks) ? sceneVisibilityTr<STATS, kMasks>(s, shadow, cnt, tr, draw)
                                                        : sceneVisibility<STATS, MESHL>(s, shadow, cnt);
                tm.mark(3);
                if (!visible) {
                    if (!shadow_catcher) return;
                    tr = S3(0.f);  // pathtracer_mis.zig:251-257
                }
                const float      lr = h.smp.sample1D();  // light.zig:127 rs.stochastic_r: drawn only for unoccluded samples
                const f3         radiance = (ZYG_EXPERIMENT_NEE_SKIP & 8) ? S3(lr) : materialRadiance<MESHL>(s, lm, ls.wi, trafo, light.prop, light.part, false, ls.u, ls.v, lr, sf.p);
                if (1 == PHASE) {  // the closure's half of this sample happens in wf_scatter<.., 2>
                    if (stored < kMaxNeeStored) {
                        const size_t at = size_t(3u * stored) * batchPaths() + ps;
                        P.nee_rec[at] = f4(ls.wi, ls.pdf * pick.pdf);
                        P.nee_rec[at + batchPaths()] = f4(radiance, __uint_as_float(pk));
                        P.nee_rec[at + 2 * size_t(batchPaths())] = f4(tr, light.shadow_catcher_light ? 1.f : 0.f);
                        stored += 1;
                    } else {
                        overflow = true;
                    }
                    return;
                }
                BxdfResult       bx;
                if (ZYG_EXPERIMENT_NEE_SKIP & 2) bx.reflection = ls.wi, bx.pdf = ls.pdf;
                else if (kRebuild) {
                    uint32_t again = slot;
                    pin(again);  // (the loads behind it cannot be merged with the first build's)
                    const Surface se = loadSurface<POOL, MESHL>(P, again, h.flags, record);
                    bx = se.ms.template evaluate<SUBST, POOL, MESHL>(s, ls.wi, POOL && max_splits > 1);
                } else if (MESHL) {  // (pin: the closure's light-independent terms are computed here, not in front of the light loop)
                    MaterialSample mj = ms;
                    pin(mj.wo);
                    pin(mj.n);
                    bx = mj.template evaluate<SUBST, POOL, MESHL>(s, ls.wi, POOL && max_splits > 1);
                } else bx = ms.template evaluate<SUBST, POOL, MESHL>(s, ls.wi, POOL && max_splits > 1);
                nanTrace(P, ls.wi, 10);
                nanTrace(P, ls.pdf, 11);
                nanTrace(P, radiance, 12);
                nanTrace(P, bx.reflection, 13);
                nanTrace(P, bx.pdf, 14);
                nanTrace(P, pick.pdf, 15);
                if (mis) {
                    const float light_pdf = ls.pdf * pick.pdf;
                    const float weight = predividedPowerHeuristic(light_pdf, bx.pdf);
                    const f3    unocc = S3(weight) * radiance * bx.reflection;
                    if (catcher_scene) unoccluded = unoccluded + unocc;
                    occluded = occluded + ((POOL || kMasks) ? tr * unocc : unocc);  // pathtracer_mis.zig:267-270 (tr = 1 without thin glass)
                } else {
                    const float weight = 1.f / (pick.pdf * ls.pdf);
                    occluded = occluded + S3(weight) * (((POOL || kMasks) ? tr : S3(1.f)) * radiance * bx.reflection);
                }
                tm.mark(4);
            };

            // A shape's sampleTo draws all of its light samples before the first shadow test (rectangle.zig:321-352,
            // triangle_mesh.zig:513-592); instead of buffering them, a copy of the sampler replays those draws while
            // the path's sampler runs ahead to where the post-visibility draws continue.
            if (MESHL) {
                // The instantiations that know every light kind: `connect` (the shadow ray's walk through every shape, the light's
                // radiance, the complete closure's evaluation) is expanded at ONE place, behind a switch that only makes the light
                // sample.  With one expansion per light kind (ten of them) wf_scatter<.., MESHL, POOL> was 1.16 MB of code with 886
                // spilled registers: every wave's instruction stream missed the 64 KB instruction cache (profiles/r04a_*).
                // Phase 1 decides how many samples the light takes and moves the path's sampler past the draws sampleTo makes before
                // the first shadow test; phase 2 makes sample j from the replayed draws.  The per-light set-up (SphQuad, cone frame,
                // chord, portal window) is made again in phase 2 instead of being carried across `connect`, whose registers it
                // would add to: a light takes one sample but for the emittance's num_samples.
                constexpr auto has = [](uint32_t kind) { return 0 != (KINDS & (1u << kind)); };  // compiled into this instantiation
                uint32_t lk = LK_NONE, n = 0;
                Sampler  replay = h.smp;
                float    mesh_r = 0.f;
                auto     advance2D = [&](uint32_t count) {
                    for (uint32_t j = 0; j < count; ++j) {
                        float a, b;
                        h.smp.sample2D(a, b);
                    }
                };
                if (ZYG_EXPERIMENT_NEE_SKIP & 4) {
                    lk = LK_RECTANGLE;
                    n = 1;
                } else if (has(LK_PORTAL) && ZYGPU_LIGHT_PORTAL_IMAGE == light.light_class) {
                    // Light.propSamplePortalTo (light.zig:217-240) -> Rectangle.samplePortalTo (rectangle.zig:458-513): the window the
                    // point sees the portal through, then the map's windowed distribution; no window, no draw
                    Bounds2 bounds;
                    if (portalImageBounds(sf.p, trafo, bounds)) {
                        lk = LK_PORTAL;
                        n = numSamples(light.num_samples, sample_threshold);
                        advance2D(n);
                    }
                } else if ((has(LK_IMAGE_FLAT) || has(LK_IMAGE_SPHERE) || has(LK_IMAGE_CANOPY)) && ZYGPU_LIGHT_PROP_IMAGE == light.light_class) {
                    // Light.propSampleMaterialTo (light.zig:191-215): the emission map's distribution picks the direction
                    if (has(LK_IMAGE_FLAT) && (ZYGPU_SHAPE_DOME == light_shape || ZYGPU_SHAPE_RECTANGLE == light_shape || ZYGPU_SHAPE_DISK == light_shape)) {
                        lk = LK_IMAGE_FLAT;
                        n = numSamples(light.num_samples, sample_threshold);
                        advance2D(n);
                    } else {
                        lk = (has(LK_IMAGE_SPHERE) && ZYGPU_SHAPE_SPHERE == light_shape) ? LK_IMAGE_SPHERE : LK_IMAGE_CANOPY;  // one sample, drawn from the path's sampler
                        n = 1;
                    }
                } else if (has(LK_MESH) && kNull != light.sampler) {
                    const DMeshLight&      ml = s.mesh_lights[light.sampler];
                    const MeshLightSampler msr = MeshLightSampler::init(sf.p, ms.geo_n, trafo);
                    mesh_r = h.smp.sample1D();
                    const LightTree ptree{s, ml.tree};
                    replay = h.smp;
                    lk = LK_MESH;
                    n = ptree.randomLight(notimer, msr.op, msr.on, ms.translucent, mesh_r, sample_threshold, [](uint32_t, LightPick) {});
                    advance2D(n);
                } else if (has(LK_INFINITE) && distant) {
                    lk = LK_INFINITE;
                    n = 1;
                } else if (has(LK_SPHERE) && ZYGPU_SHAPE_SPHERE == light_shape) {
                    const uint32_t      ns = numSamples(light.num_samples, sample_threshold);
                    const SphereSampler ss = SphereSampler::init(sf.p, trafo, ns);
                    if (ss.valid) {  // sphere.zig:331-337: a point inside the sphere takes no sample and no draw
                        lk = LK_SPHERE;
                        n = ns;
                        advance2D(n);
                    }
                } else if (has(LK_DISK) && ZYGPU_SHAPE_DISK == light_shape) {
                    const uint32_t    ns = numSamples(light.num_samples, sample_threshold);
                    const DiskSampler ds = DiskSampler::init(sf.p, trafo, ns);
                    if (ds.valid) {  // disk.zig:273-276: no sample, no draw
                        // a sample's second draw depends on its first: the path's sampler runs the sampling once for its
                        // draws alone, the copy once more for the samples
                        lk = LK_DISK;
                        n = ns;
                        for (uint32_t j = 0; j < n; ++j) {
                            SampleTo ls;
                            ds.sample(h.smp, sf.p, ms.geo_n, trafo, 0 != light.two_sided, ms.translucent, ls);
                        }
                    }
                } else if (has(LK_RECTANGLE)) {
                    lk = LK_RECTANGLE;
                    n = MULTI ? numSamples(light.num_samples, sample_threshold) : 1u;  // shape_sampler.zig:35-41
                    if (MULTI) advance2D(n);
                }
                for (uint32_t j = 0; j < n; ++j) {
                    SampleTo ls;
                    bool     got = false;
                    f3       pj = sf.p, nj = ms.geo_n;  // (pin: the light kinds' set-up stays inside its case)
                    pin(pj);
                    pin(nj);
                    switch (lk) {
                        case LK_PORTAL: if (has(LK_PORTAL)) {
                            const DPortalLight& pl = s.portal_lights[light.sampler];
                            const Trafo         dome = loadTrafo(s, s.lights[pl.light_link].prop);  // (the host refuses an animated link)
                            Bounds2             bounds;
                            portalImageBounds(pj, trafo, bounds);
                            float r0, r1;
                            replay.sample2D(r0, r1);
                            got = rectangleSamplePortal(pl, dome, bounds, r0, r1, float(n), pj, nj, trafo, ms.translucent, ls);
                            break;
                        }
                        case LK_IMAGE_FLAT: if (has(LK_IMAGE_FLAT)) {
                            const DImageLight& il = s.image_lights[light.sampler];
                            float              r0, r1;
                            replay.sample2D(r0, r1);
                            got = ZYGPU_SHAPE_DOME == light_shape
                                      ? domeSampleMaterial(il, r0, r1, float(n), nj, trafo, ms.translucent, ls)
                                  : ZYGPU_SHAPE_DISK == light_shape
                                      ? diskSampleMaterial(il, r0, r1, float(n), pj, nj, trafo, 0 != light.two_sided, ms.translucent, ls)
                                      : rectangleSampleMaterial(il, r0, r1, float(n), pj, nj, trafo, 0 != light.two_sided, ms.translucent, ls);
                            break;
                        }
                        case LK_IMAGE_SPHERE: if (has(LK_IMAGE_SPHERE)) got = sphereSampleMaterialTo(s.image_lights[light.sampler], pj, nj, trafo, ms.translucent, h.smp, ls); break;
                        case LK_IMAGE_CANOPY: if (has(LK_IMAGE_CANOPY)) got = canopySampleMaterialTo(s.image_lights[light.sampler], nj, trafo, ms.translucent, h.smp, ls); break;
                        case LK_MESH: if (has(LK_MESH)) {
                            const DMeshLight&      ml = s.mesh_lights[light.sampler];
                            const DMesh&           mesh = s.meshes[ml.mesh];
                            const MeshLightSampler msr = MeshLightSampler::init(pj, nj, trafo);
                            const LightTree        ptree{s, ml.tree};
                            LightPick              tri{};
                            // the primitive tree's picks again, keeping the j-th (they are a function of (point, normal, r) alone)
                            ptree.randomLight(notimer, msr.op, msr.on, ms.translucent, mesh_r, sample_threshold, [&](uint32_t k, LightPick t) {
                                if (k == j) tri = t;
                            });
                            float r0, r1;
                            replay.sample2D(r0, r1);
                            got = msr.sample(mesh, ml, tri, r0, r1, pj, nj, trafo, 0 != light.two_sided, ms.translucent, ls);
                            break;
                        }
                        case LK_INFINITE:
                            if (has(LK_INFINITE)) got = ZYGPU_SHAPE_DOME == light_shape     ? domeSampleTo(nj, ms.translucent, h.smp, ls)
                                  : ZYGPU_SHAPE_CANOPY == light_shape ? canopySampleTo(nj, trafo, ms.translucent, h.smp, ls)
                                                                      : distantSampleTo(nj, trafo, ms.translucent, h.smp, ls);
                            break;
                        case LK_SPHERE: if (has(LK_SPHERE)) {
                            const SphereSampler ss = SphereSampler::init(pj, trafo, n);
                            float               s0, s1;
                            replay.sample2D(s0, s1);
                            got = ss.sample(s0, s1, pj, nj, trafo, ms.translucent, ls);
                            break;
                        }
                        case LK_DISK: if (has(LK_DISK)) {
                            const DiskSampler ds = DiskSampler::init(pj, trafo, n);
                            got = ds.sample(replay, pj, nj, trafo, 0 != light.two_sided, ms.translucent, ls);
                            break;
                        }
                        default: if (has(LK_RECTANGLE)) {  // LK_RECTANGLE
                            const RectangleSampler rs = RectangleSampler::init(pj, trafo, n);
                            float                  u0, u1;
                            if (MULTI) replay.sample2D(u0, u1);
                            else h.smp.sample2D(u0, u1);
                            got = rs.sample(u0, u1, pj, nj, trafo, 0 != light.two_sided, ms.translucent, ls);
                            break;
                        }
                    }
                    if (got) connect(ls);
                }
            } else {  // the instantiations whose lights are all Rectangles
                const uint32_t         n = MULTI ? numSamples(light.num_samples, sample_threshold) : 1u;  // shape_sampler.zig:35-41
                const RectangleSampler rs = RectangleSampler::init(sf.p, trafo, n);
                Sampler                replay = h.smp;
                if (MULTI) {
                    for (uint32_t j = 0; j < n; ++j) {
                        float a, b;
                        h.smp.sample2D(a, b);
                    }
                }
                for (uint32_t j = 0; j < n; ++j) {
                    float u0, u1;
                    if (MULTI) replay.sample2D(u0, u1);
                    else h.smp.sample2D(u0, u1);
                    SampleTo ls;
                    if (rs.sample(u0, u1, sf.p, ms.geo_n, trafo, 0 != light.two_sided, ms.translucent, ls)) connect(ls);
                }
            }
            if (1 == PHASE) continue;
            if (shadow_catcher && light.shadow_catcher_light) {  // pathtracer_mis.zig:274-276
                catcher_occluded = catcher_occluded + occluded;
                catcher_unoccluded = catcher_unoccluded + unoccluded;
            } else {
                nee = nee + occluded;
            }
        }
        if (preload_sum) {  // ivalueAdd's arithmetic on the sum read above (the direct sum shares its four words with emission.z: .x)
            const f3 value = throughput * nee;
            if (!(0.f == value.x && 0.f == value.y && 0.f == value.z))
                *sum_at = sum_is_direct ? make_float4(sum_old.x, sum_old.y + value.x, sum_old.z + value.y, sum_old.w + value.z)
                                        : f4(xyz4(sum_old) + value, sum_old.w);
        } else if (1 != PHASE) ivalueAdd<!POOL>(P, ps, (POOL ? throughput * S3(split_weight) : throughput) * nee, h.depth_surface, 1u, false, false);
        tm.mark(5);
    }
    if (1 == PHASE) {
        // a vertex with more light samples than the record holds is left as it was (nothing of it has been stored) and goes through
        // the unsplit kernel; every other vertex hands its sampler on, behind sampleLights' draws
        if (overflow) {
            P.nee_count[ps] = kNeeOverflow;
            out.push = true;
        } else {
            P.nee_count[ps] = stored;
            if (stored || texture_draws || (ZYGPU_INTEGRATOR_PT != kind && ms.can_evaluate)) storeSamplers(P, ps, h);
        }
        return;
    }
    if (catcher_scene) {  // pathtracer_mis.zig:108-115 (whether or not the material could be evaluated)
        h.flags = shadow_catcher ? (h.flags | ST_FROM_CATCHER) : (h.flags & ~uint32_t(ST_FROM_CATCHER));
        if (shadow_catcher) {
            P.pool.sc_occ[slot] = f4(catcher_occluded);
            P.pool.sc_unocc[slot] = f4(catcher_unoccluded);
            h.flags |= ST_CATCHER_IN_CAMERA;
        }
    }

    // ---- bxdf sample + path extension (pathtracer_mis.zig:123-166)
    if (kRebuild && ZYGPU_INTEGRATOR_PT != kind && ms.can_evaluate) {  // (the light loop ran: the closure it let go of)
        uint32_t again = slot;
        pin(again);
        sf = loadSurface<POOL, MESHL>(P, again, h.flags, record);
    }
    if (1 == ZYG_EXPERIMENT_SCATTER_PART) {
        storeHead<POOL>(P, slot, h);
        return;
    }
    BxdfSample bs;
    if (POOL) {
        // mat_sample.sample(sampler, max_splits, ..): Glass splits into reflection and refraction while the sample has
        // vertices left (Pool.maxSplits, vertex.zig:305-308; PathtracerMIS only, pathtracer.zig:73, pathtracer_dl.zig:82)
        BxdfSample     bs1;
        uint32_t       count;
        const uint32_t path_count = __float_as_uint(P.pool.vorigin.get(slot).w);
        if (ZYGPU_MAT_GLASS == ms.kind) {
            // rs.wavelength (vertex.zig:167): only a dispersive glass looks at it
            const float path_wavelength = 0.f != ms.abbe ? __uint_as_float(P.pool.med1.get(slot).w) : 0.f;
            count = ms.glassSample(s, h.smp, max_splits > 1, bs, bs1, path_wavelength);
        } else {
            bs.wavelength = 0.f;
            count = ms.template sample<SUBST, MESHL>(s, h.smp, bs) ? 1u : 0u;
        }
        h.smp.incrementPadding();  // after the children, none of which draws (pathtracer_mis.zig:165)
        storeSamplers(P, ps, h);
        if (0 == count) {
            terminateVertex(P, ps, S3(0.f), split_weight, h.flags, slot);  // vertex.throughput = 0 (pathtracer_mis.zig:127-129)
        } else {
            Mediums md;
            md.load(P, slot);
            const uint32_t st = P.pool.pstate.get(ps);
            const uint32_t first = (st >> 8) & 255u;
            P.pool.pstate[ps] = st + (count << 8) + (count - 1u);  // children created; vertices in flight: -1 + count
            const float  parent_pdf = sf.parent_pdf;
            const float4 parent_vgeo_n = P.pool.vgeo_n[slot], parent_vorigin = P.pool.vorigin[slot];
            for (uint32_t c = 0; c < count; ++c) {
                const BxdfSample& b = c ? bs1 : bs;
                const uint32_t    cv = (nxt * kPoolVertices + first + c) * batchPaths() + ps;
                PathHead          ch = h;
                // Vertex.State.update (vertex.zig:30-43): a .Straight event (thin glass) leaves the state alone
                if (kScatterSpecular == b.scattering) {
                    ch.flags |= ST_SPECULAR;
                    ch.flags = (0.f == b.reg_alpha) ? (ch.flags | ST_SINGULAR) : (ch.flags & ~uint32_t(ST_SINGULAR));
                    if (ch.flags & ST_PRIMARY) ch.flags |= ST_STARTED_SPECULAR;
                } else if (!b.straight) {
                    ch.flags &= ~uint32_t(ST_SPECULAR | ST_SINGULAR | ST_PRIMARY);
                }
                float bxdf_pdf = parent_pdf;
                if (mis && !b.straight) {  // pathtracer_mis.zig:141-148: not for .Straight
                    ch.flags = ms.translucent ? (ch.flags | ST_TRANSLUCENT) : (ch.flags & ~uint32_t(ST_TRANSLUCENT));
                    ch.vdepth = h.depth_surface;
                    bxdf_pdf = b.pdf;
                }
                const f3 new_origin = surfaceOffsetP(sf.p, sf.frag_geo_n, b.wi);
                ch.depth_surface = h.depth_surface + 1;
                Mediums cm = md;
                if (0.f == cm.wavelength) cm.wavelength = b.wavelength;  // pathtracer_mis.zig:155-157
                if (b.transmission) {
                    // Vertex.interfaceChange (vertex.zig:94-108): leaving if the new direction is on the side of the
                    // fragment's geometric normal
                    const uint32_t  hit_prop = P.pool.hit_prop[slot];
                    const ZygpuProp hp = s.props[hit_prop];
                    const uint32_t  part =
                        ZYGPU_SHAPE_TRIANGLE_MESH == hp.shape ? uint32_t(s.meshes[hp.mesh].triangle_parts[P.pool.hit.get(slot).w]) : 0u;
                    if (dot3(sf.frag_geo_n, b.wi) > 0.f) cm.remove(hit_prop, part);
                    else cm.push(hit_prop, part);
                } else if (!b.straight) {
                    ch.flags &= ~uint32_t(ST_TRANSPARENT);  // transparent && (Transmission || Straight)
                }
                P.pool.ray_o[cv] = f4(new_origin, kRayMaxT);
                P.pool.ray_d[cv] = f4(b.wi, bxdf_pdf);
                P.pool.thr[cv] = f4(throughput * (b.reflection / S3(b.pdf)), split_weight * b.split_weight);
                // vertex.origin / geo_n move with every event but .Straight (pathtracer_mis.zig:141-148, pathtracer.zig:88-90)
                P.pool.vorigin[cv] = f4(b.straight ? xyz4(parent_vorigin) : sf.p, __uint_as_float(path_count * count));
                P.pool.vgeo_n[cv] = (mis && !b.straight) ? f4(ms.geo_n, b.reg_alpha) : parent_vgeo_n;
                P.pool.vmeta[cv] = packMeta(ch);
                if (catcher_scene) {  // next_vertex.* = vertex.* (pathtracer_mis.zig:133)
                    P.pool.sc_occ[cv] = P.pool.sc_occ[slot];
                    P.pool.sc_unocc[cv] = P.pool.sc_unocc[slot];
                    P.pool.sc_emis[cv] = P.pool.sc_emis[slot];
                }
                cm.store(P, cv);
                child[c] = cv;
                child_round[c] = first + c;
                child_back[c] = entersMeshBounds(new_origin, b.wi);
            }
        }
    } else if (!ms.template sample<SUBST, MESHL>(s, h.smp, bs)) {
        finalize(P, slot, S3(0.f), h.flags, h.smp.rng);
    } else {
        // Vertex.State.update (vertex.zig:30-43); every sample in scope is a reflection (never .Straight)
        if (kScatterSpecular == bs.scattering) {
            h.flags |= ST_SPECULAR;
            h.flags = (0.f == bs.reg_alpha) ? (h.flags | ST_SINGULAR) : (h.flags & ~uint32_t(ST_SINGULAR));
            if (h.flags & ST_PRIMARY) h.flags |= ST_STARTED_SPECULAR;
        } else {
            h.flags &= ~uint32_t(ST_SPECULAR | ST_SINGULAR | ST_PRIMARY);
        }
        const bool packed_prev = 0 != P.packed_prev;
        float      bxdf_pdf;
        if (mis) {
            h.flags = ms.translucent ? (h.flags | ST_TRANSLUCENT) : (h.flags & ~uint32_t(ST_TRANSLUCENT));
            h.vdepth = h.depth_surface;
            bxdf_pdf = bs.pdf;
#if !ZYG_EXPERIMENT_NO_LINE_B
            if (!packed_prev) P.pool.vgeo_n[slot] = f4(ms.geo_n, bs.reg_alpha);
#endif
        } else {
            bxdf_pdf = P.pool.ray_d.get(slot).w;  // the other integrators pass the camera ray's 0 on: read only there
        }
        throughput = throughput * (bs.reflection / S3(bs.pdf));
        nanTrace(P, bs.reflection, 16);
        nanTrace(P, bs.pdf, 17);
        nanTrace(P, bs.wi, 18);
        nanTrace(P, sf.p, 19);
        nanTrace(P, throughput, 20);
        const f3 new_origin = surfaceOffsetP(sf.p, sf.frag_geo_n, bs.wi);
        h.depth_surface += 1;
        h.flags &= ~uint32_t(ST_TRANSPARENT);  // transparent && (Transmission || Straight): a reflection clears it
        h.smp.incrementPadding();

        P.pool.ray_o[slot] = f4(new_origin, kRayMaxT);
        P.pool.ray_d[slot] = f4(bs.wi, bxdf_pdf);
        back = entersMeshBounds(new_origin, bs.wi);
        if (packed_prev) {  // (only PathtracerMIS looks at geo_n and the roughness: the other integrators leave them behind)
            const PrevVertex prev{sf.p, mis ? ms.geo_n : S3(0.f), mis ? bs.reg_alpha : 0.f};
            P.pool.thr[slot] = f4(throughput, prev.alpha);
            storeHead(P, slot, h, &prev);
            storePrevTail(P, slot, prev);
        } else {
            P.pool.thr[slot] = f4(throughput);
#if !ZYG_EXPERIMENT_NO_LINE_B
            P.pool.vorigin[slot] = f4(sf.p);
#endif
            storeHead(P, slot, h);
        }
        push = true;
    }
    tm.mark(6);
}

// Occupancy: the plain instantiations 3 waves per SIMD, those with Substitute / multi-sample / mesh-light code 2; the POOL instantiation
// that knows everything (every light kind, the complete closures, Glass with its split paths) ONE: with 256 registers it spilled
// 554 of them and its time was the scratch traffic (wf_scatter 3.3 KB of HBM traffic per scattered vertex against ~0.7 KB of state,
// profiles/r04e_cfg4_*), with 512 it spills 147 (relaxed units) and runs faster on half the waves: cornell_glass 115.3 -> 130.0
// Msamples/s, the outdoor scene 218.3 -> 223.6 (gpurun_out/r04f, one call).
#ifndef ZYG_SCATTER_POOL_FULL_WAVES
#define ZYG_SCATTER_POOL_FULL_WAVES 1
#endif
#ifndef ZYG_LIGHT_WAVES
#define ZYG_LIGHT_WAVES 2  // wf_light: waves per SIMD
#endif
#ifndef ZYG_LIGHT_SKY_WAVES
#define ZYG_LIGHT_SKY_WAVES 3  // ... of the instantiation for a sun and a sky map (217 registers without a spill; at 168: 40 spilled, and faster)
#endif
#ifndef ZYG_SCATTER_COMPLETE_WAVES
#define ZYG_SCATTER_COMPLETE_WAVES 2  // the complete closures outside POOL scenes (textured / coated Substitute, mesh lights)
#endif
#ifndef ZYG_SCATTER_SORT
#define ZYG_SCATTER_SORT 0  // 1: the complete instantiations outside POOL scenes deal their vertices prop by prop (dealByProp); 3: the plain ones too
#endif
#ifndef ZYG_SCATTER_EVAL_WAVES
#define ZYG_SCATTER_EVAL_WAVES 2  // wf_scatter<.., PHASE 2>
#endif
// The light half of the POOL scenes' complete wf_scatter (scatterVertex PHASE 1): one lane per alive vertex, no closure, no queue
// append but for the vertices that overflow the sample record (to Queues::shadow, which the unsplit kernel then reads).
// KINDS: the light kinds compiled in.  All of them: 256 registers with 87 spilled; Rectangle lights only: 256 with 6 (cornell_glass
// 140 -> 149 Msamples/s); a Sky entity's sun and sky map: 217 with none, run at three waves per SIMD with 40 spilled (the outdoor scene
// at 3840x2160 361 -> 367 at two waves -> 390 at three; the Rectangle one at three: 137): gpurun_out/r04r, one call.
template <bool MULTI, bool MASKS, uint32_t KINDS = kAllLightKinds>
__global__ void __launch_bounds__(256) ZYG_WAVES(kSkyLightKinds == KINDS ? ZYG_LIGHT_SKY_WAVES : ZYG_LIGHT_WAVES) wf_light(Params P) {
    if (blockIdx.x * blockDim.x >= aliveBound(P, P.parity)) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at;
    bool           has_entry;
    if (!aliveOpen(P, P.parity, i, at, has_entry)) return;  // whole blocks take part in queuePush
    Counters           cnt{0, 0, 0};
    unsigned long long shadow_rays = 0;
    SectionTimer       tm;
    tm.start(true);
    ScatterOut out;
    uint32_t   slot = 0;
    if (has_entry) {
        slot = P.alive_in[at];
        scatterVertex<false, true, MULTI, true, true, MASKS, false, false, 1, KINDS>(P, c_scene, slot, out, cnt, shadow_rays, tm, nullptr);
    }
    queuePush(P.q.shadow, &P.ctl->shadow_count[0], P.queue_capacity, out.push, slot);
    tm.flush(P.ctl, 16);
}

// KINDS: the light kinds the scene has (the complete instantiations of scenes whose lights are all plain Rectangles: a scene is on the
// complete kernels for its textures, shapes or keyframes as often as for its lights)
template <bool STATS, bool SUBST, bool MULTI, bool MESHL, bool POOL, bool MASKS = STATS, bool UTLAS = false, bool WIDE = false, int PHASE = 0, uint32_t KINDS = kAllLightKinds>
__global__ void __launch_bounds__(256) ZYG_WAVES(2 == PHASE ? ZYG_SCATTER_EVAL_WAVES : (MESHL && POOL && !STATS) ? ZYG_SCATTER_POOL_FULL_WAVES : (MESHL && !STATS) ? ZYG_SCATTER_COMPLETE_WAVES : (SUBST || MULTI || MESHL) ? ZYG_SCATTER_BIG_WAVES : ZYG_SCATTER_WAVES) wf_scatter(Params P) {
    static_assert(!UTLAS || (!STATS && !MESHL && !POOL && !MASKS), "the prop table holds Rectangles and meshes only and counts nothing");
    static_assert(0 == PHASE || (2 == PHASE && POOL && !STATS), "the split next-event estimation: the POOL scenes' instantiations");
    const uint32_t     nxt = P.parity ^ 1u;
    if (blockIdx.x * blockDim.x >= aliveBound(P, P.parity)) return;
    const uint32_t     i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t           at, alive_end = 0;  // alive_end 1: an entry of the alive queue's back end (Params::alive_ends): wf_light has not seen the vertex
    bool               has_entry;
    if (POOL && P.scatter_source) {  // wf_light's overflow queue: the vertices that took more light samples than its record holds
        const uint32_t n = P.ctl->shadow_count[0].n[0];
        if (blockIdx.x * blockDim.x >= n) return;
        at = i;
        has_entry = i < n;
    } else
    if (!aliveOpen(P, P.parity, i, at, has_entry, &alive_end)) return;  // whole blocks take part in queuePush
#if ZYG_SCATTER_SORT
    if (!POOL && !STATS && ((ZYG_SCATTER_SORT & 2) || MESHL)) dealByProp<false>(P, at, has_entry);
#endif
    Counters           cnt{0, 0, 0};
    unsigned long long shadow_rays = 0;
    const DScene& s = c_scene;
    uint32_t slot = 0;
    SectionTimer tm;  // -DZYG_PROFILE_SECTIONS builds: Control::sections[16..]
    tm.start(true);
    ScatterOut out;
#if 2 == ZYG_EXPERIMENT_SKELETON  // timing experiment (wrong images): the same bytes, but 8 lanes read / write one path's whole line
    {
        constexpr uint32_t kRow = 9;  // float4 per path row in LDS: the 128 B line + 16 B of padding
        __shared__ float4 rows[256 * kRow];
        __shared__ float4 hits[256 * 2];
        const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
        slot = has_entry ? P.alive_in[at] : 0u;
        float4* my_rows = rows + wave * 64 * kRow;
        float4* my_hits = hits + wave * 64 * 2;
        for (uint32_t j = 0; j < 8; ++j) {  // line A: lanes 8k .. 8k+7 load path 8j + k's eight records
            const uint32_t src = 8u * j + (lane >> 3);
            const uint32_t s2 = __shfl(slot, src);
            const bool     ok = 0 != __shfl(has_entry ? 1 : 0, src);
            float4         v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(pathBlock(P, s2) + (lane & 7u) * 16u);
            my_rows[src * kRow + (lane & 7u)] = v;
        }
        for (uint32_t j = 0; j < 2; ++j) {  // hit block: 2 lanes per path
            const uint32_t src = 32u * j + (lane >> 1);
            const uint32_t s2 = __shfl(slot, src);
            const bool     ok = 0 != __shfl(has_entry ? 1 : 0, src);
            float4         v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(P.pool.hit.base + size_t(s2) * ZYG_HIT_STRIDE + (lane & 1u) * 16u);
            my_hits[src * 2 + (lane & 1u)] = v;
        }
        __builtin_amdgcn_wave_barrier();
        float4* r = my_rows + lane * kRow;
        const float4 hr = my_hits[lane * 2], hp = my_hits[lane * 2 + 1];
        float4 ind = r[6];
        ind.x += r[5].x * hr.x + hp.x;
        r[6] = ind;
        uint4 q0 = reinterpret_cast<uint4*>(r)[0];
        q0.y += 8u;  // depth_surface + 1
        reinterpret_cast<uint4*>(r)[0] = q0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t src = 8u * j + (lane >> 3);
            const uint32_t s2 = __shfl(slot, src);
            const bool     ok = 0 != __shfl(has_entry ? 1 : 0, src);
            if (ok) *reinterpret_cast<float4*>(pathBlock(P, s2) + (lane & 7u) * 16u) = my_rows[src * kRow + (lane & 7u)];
        }
        if (has_entry) {
            P.pool.vorigin[slot] = r[3];
            P.pool.vgeo_n[slot] = r[4];
            out.push = true;
        }
    }
#elif ZYG_EXPERIMENT_SKELETON  // timing experiment (wrong images): the kernel's path-state loads and stores with no work between them
    if (has_entry) {
        slot = P.alive_in[at];
        PathHead h;
        loadHead<POOL>(P, slot, h);
        const float4   ro = P.pool.ray_o[slot], rd = P.pool.ray_d[slot];
        const uint4    hr = P.pool.hit[slot];
        const uint32_t hp = P.pool.hit_prop[slot];
        const float4   thr = P.pool.thr[slot];
        float4         ind = P.pool.indirect[slot];
        ind.x += thr.x * __uint_as_float(hr.x) + float(hp);
        P.pool.indirect[slot] = ind;
        P.pool.ray_o[slot] = ro;
        P.pool.ray_d[slot] = rd;
        P.pool.thr[slot] = thr;
        P.pool.vorigin[slot] = ro;
        P.pool.vgeo_n[slot] = rd;
        h.depth_surface += 1;
        storeHead(P, slot, h);
        out.push = true;
    }
#else
    if (has_entry) {
        slot = (POOL && P.scatter_source) ? P.q.shadow[at] : P.alive_in[at];
        // Params::hit_q: the vertex' hit record sits at its place in the alive queue (its address does not wait for the slot)
        const HitRecord hr = (!POOL && P.hit_q) ? loadQueuedHit(P, at) : loadHitRecord(P, slot);
        scatterVertex<STATS, SUBST, MULTI, MESHL, POOL, MASKS, UTLAS, WIDE, PHASE, KINDS>(P, s, slot, out, cnt, shadow_rays, tm, &hr, POOL && 0 != alive_end);
    }
#endif
    if (POOL) {
        for (uint32_t r = 0; r < kPoolVertices; ++r) {  // child `first + c` of a sample is vertex `first + c` of the next level
            const bool c0 = r == out.child_round[0], c1 = r == out.child_round[1];
            if (r) __syncthreads();  // queuePush's shared words are reused
            queuePush(P.q.ext_pool[nxt][r], &P.ctl->ext_pool_count[nxt][r], P.queue_capacity, c0 || c1, c1 ? out.child[1] : out.child[0],
                      c1 ? out.child_back[1] : out.child_back[0]);
        }
    } else {
        queuePush(P.q.ext[nxt], &P.ctl->ext_count[nxt], P.queue_capacity, out.push, slot, out.back);
    }
    tm.mark(7);
    tm.flush(P.ctl, 16);
    if (STATS) flushCounters(P, 2, cnt, shadow_rays);
}

// ---------------------------------------------------------------- wf_textures
// The procedural textures of the tick's surviving vertices (dprocedural.hpp), launched between the shading passes and the kernels that
// build the closure, in scenes that have a material with ZYGPU_MATF_PROCEDURAL only.  One thread per entry of the alive queue (POOL:
// of the round's).  The vertex' Renderstate is rebuilt as loadSurface rebuilds it; the textures are evaluated in the order
// Material.sample reads them (substitute_material.zig:121-187, glass_material.zig:50-73), so that the sampler dimensions Mix textures
// draw are the reference's; the draws are not written back: the kernel that takes the vertex' light and bxdf samples repeats them
// (skipTextureDraws).  Compiled in one kernel unit (kernel_table.hpp); the others see the declaration.
#ifndef ZYG_TEXTURES_UNIT
#define ZYG_TEXTURES_UNIT 7  // a unit compiled with the default scheduler (build.py): max-ilp costs this kernel 8 %
#endif
template <bool POOL>
__global__ void wf_textures(Params P);

#if defined(ZYG_TU) && ZYG_TEXTURES_UNIT == ZYG_TU
}  // namespace zd
#include "dprocedural.hpp"
namespace zd {

#ifndef ZYG_TEXTURES_SORT
#define ZYG_TEXTURES_SORT 1  // 0: thread i evaluates alive entry i
#endif
#ifndef ZYG_TEXTURES_WAVES
#define ZYG_TEXTURES_WAVES 4  // waves per SIMD (127 registers, none spilled; 2 / 3 / 4 / 5 / 6 / 8: 347 / 367 / 379 / 369 / 363 / 344 Msamples/s on the procedural Cornell box)
#endif
template <bool POOL>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_TEXTURES_WAVES) wf_textures(Params P) {
    if (blockIdx.x * blockDim.x >= aliveBound(P, P.parity)) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at;
    bool           has_entry;
    if (!aliveOpen(P, P.parity, i, at, has_entry)) return;
#if ZYG_TEXTURES_SORT
    dealByProp<POOL>(P, at, has_entry);
#endif
    if (!has_entry) return;
    const DScene&  s = c_scene;
    const uint32_t slot = P.alive_in[at];
    const uint32_t ps = sampleSlot<POOL>(slot);
    PathHead       h;
    loadHead<POOL>(P, slot, h);
    TexState rs;
    uint32_t material;
    f3       wo;
    if (!POOL) {
        const HitRecord hr = P.hit_q ? loadQueuedHit(P, at) : loadHitRecord(P, slot);
        if (hr.lazy) {  // the first shading pass' two draws (scatterVertex repeats them the same way)
            (void)h.smp.sample1D();
            (void)h.smp.sample1D();
        }
        const float4 ro = P.pool.ray_o[slot], rd = P.pool.ray_d[slot];
        const Hit    hit{__uint_as_float(hr.hit.x), __uint_as_float(hr.hit.y), __uint_as_float(hr.hit.z), hr.hit.w, hr.prop};
        Ray          ray = Ray::init(xyz4(ro), xyz4(rd), 0.f, hit.t);
        ray.time = pathTime<false, true>(P, slot);
        Fragment frag;
        makeFragment<true>(s, ray, hit, frag);
        wo = -ray.direction;
        material = s.material_ids[s.props[frag.prop].parts_start + frag.part];
        rs.p = frag.p;
        rs.geo_n = frag.geo_n;
        if ((s.materials[material].flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo)) rs.geo_n = -frag.geo_n;  // vertex.zig:155-161
        rs.trafo = frag.trafo;
        rs.u = frag.uv0;
        rs.v = frag.uv1;
        rs.stochastic_r = hr.stochastic_r;
        rs.prop = frag.prop;
        rs.primitive = hit.primitive;
    } else {
        const float4 rp = P.pool.rec_p[slot], rg = P.pool.rec_gn[slot];
        const float2 uv = P.pool.rec_uv[slot];
        material = __float_as_uint(rp.w);
        rs.p = xyz4(rp);
        rs.geo_n = xyz4(rg);  // (flipped by wf_shade where the material is two-sided)
        rs.prop = P.pool.hit_prop[slot];
        rs.trafo = propTrafoAt<true>(s, rs.prop, pathTime<true, true>(P, slot));
        rs.u = uv.x;
        rs.v = uv.y;
        rs.stochastic_r = P.pool.hit_sr[slot];
        rs.primitive = P.pool.hit.get(slot).w;
    }
    const ZygpuMaterial& m = s.materials[material];
    if (!(m.flags & ZYGPU_MATF_PROCEDURAL)) return;
    {  // context.camera at rs.time (scene.propTransformationAt, context.zig:140, 213)
        const ZygpuViewDesc& v = c_view.v;
        if (v.camera_num_keyframes > 1) {
            rs.camera = cameraTrafoSince(v, P.pool.time[ps]);
        } else {
            rs.camera.r0 = mk3(v.camera_trafo.r[0][0], v.camera_trafo.r[0][1], v.camera_trafo.r[0][2]);
            rs.camera.r1 = mk3(v.camera_trafo.r[1][0], v.camera_trafo.r[1][1], v.camera_trafo.r[1][2]);
            rs.camera.r2 = mk3(v.camera_trafo.r[2][0], v.camera_trafo.r[2][1], v.camera_trafo.r[2][2]);
            rs.camera.scale = mk3(v.camera_trafo.r[0][3], v.camera_trafo.r[1][3], v.camera_trafo.r[2][3]);
            rs.camera.position = mk3(v.camera_trafo.position[0], v.camera_trafo.position[1], v.camera_trafo.position[2]);
        }
    }
    rs.smp = &h.smp;
    rs.draws = 0;
    constexpr int kTop = ZYGPU_PROCEDURAL_MAX_DEPTH - 1;
    // the block's values are collected here and stored whole (a block written in part is fetched first); every index is a constant once
    // the lambdas are inlined, so the array lives in registers
    float out[kTexPlanes];
    static_assert(16 == kTexPlanes, "the block is stored as four float4");
    for (uint32_t k = 0; k < kTexPlanes; ++k) out[k] = 0.f;
    auto procedural = [](const ZygpuTexture& t) { return kNull != t.image && 0 != (t.sampler & ZYGPU_TEX_PROCEDURAL); };
    // ts.sample2D_1 of tex[k]: the value where the kernel needs it itself (coating_scale), stored where the texture is procedural
    auto weight = [&](uint32_t k, float uniform) __attribute__((always_inline)) {
        const ZygpuTexture& t = m.tex[k];
        if (!procedural(t)) return kNull == t.image ? uniform : texture1<0>(s, t, &uniform, rs);
        const float value = procedural1<kTop>(s, t.image, t.sampler, rs);
        out[kTexWeight + k] = value;
        return value;
    };
    auto weightIfProcedural = [&](uint32_t k) __attribute__((always_inline)) {
        if (procedural(m.tex[k])) out[kTexWeight + k] = procedural1<kTop>(s, m.tex[k].image, m.tex[k].sampler, rs);
    };
    auto normalIfProcedural = [&](uint32_t k, uint32_t at_plane) __attribute__((always_inline)) {
        if (!procedural(m.tex[k])) return;
        const float2 n = procedural2<kTop>(s, m.tex[k].image, m.tex[k].sampler, rs);
        out[at_plane] = n.x;
        out[at_plane + 1u] = n.y;
    };
    if (ZYGPU_MAT_GLASS == m.kind) {  // glass_material.zig:50-73: roughness, specular, the normal map
        weightIfProcedural(ZYGPU_STEX_ROUGHNESS);
        weightIfProcedural(ZYGPU_STEX_SPECULAR);
        normalIfProcedural(ZYGPU_STEX_NORMAL, kTexNormal);
    } else {  // substitute_material.zig:121-187
        if (kNull != m.color_image && (m.color_sampler & ZYGPU_TEX_PROCEDURAL)) {
            const f3 c = procedural3<kTop>(s, m.color_image, m.color_sampler, rs);
            out[kTexColor] = c.x;
            out[kTexColor + 1u] = c.y;
            out[kTexColor + 2u] = c.z;
        }
        weightIfProcedural(ZYGPU_STEX_ROUGHNESS);
        weightIfProcedural(ZYGPU_STEX_METALLIC);
        weightIfProcedural(ZYGPU_STEX_SPECULAR);
        const float coating_scale = weight(ZYGPU_STEX_COATING_SCALE, m.coating_scale);
        weightIfProcedural(ZYGPU_STEX_TRANSLUCENCY);
        normalIfProcedural(ZYGPU_STEX_NORMAL, kTexNormal);
        if (coating_scale * m.coating_thickness > 0.f) {
            const ZygpuTexture &nm = m.tex[ZYGPU_STEX_NORMAL], &cnm = m.tex[ZYGPU_STEX_COATING_NORMAL];
            const bool same = nm.image == cnm.image && (kNull == nm.image || (nm.sampler == cnm.sampler && nm.scale[0] == cnm.scale[0] && nm.scale[1] == cnm.scale[1]));
            if (!same) normalIfProcedural(ZYGPU_STEX_COATING_NORMAL, kTexCoatingNormal);
            weightIfProcedural(ZYGPU_STEX_COATING_ROUGHNESS);
        }
        weightIfProcedural(ZYGPU_STEX_ROTATION);
        weightIfProcedural(ZYGPU_STEX_FLAKES_COVERAGE);
    }
    out[kTexDraws] = __uint_as_float(rs.draws);
    float4* block = reinterpret_cast<float4*>(texBlock(P, slot));
    for (uint32_t k = 0; k < kTexPlanes / 4u; ++k) block[k] = make_float4(out[4u * k], out[4u * k + 1u], out[4u * k + 2u], out[4u * k + 3u]);
}
#endif

// ---------------------------------------------------------------- wf_tick
// wf_scatter for vertex k and wf_trace_shade for vertex k + 1 of the same path in one launch (option "fuse_tick"; scenes
// walked through the prop tables): the new ray, the path's line and the hit are still in the caches when the closest hit and
// the first shading pass use them, so per vertex one write and one read of the 128 B state line and of the hit block do not
// reach HBM, and a tick is one launch (plus wf_shade's rare second pass) instead of two.  It reads the previous tick's alive
// queue and appends to this tick's; the host zeroes this tick's counters before the launch.
template <bool SUBST, bool MULTI, bool WIDE>
__global__ void __launch_bounds__(256) ZYG_WAVES((SUBST || MULTI) ? ZYG_SCATTER_BIG_WAVES : ZYG_SCATTER_WAVES) wf_tick(Params P) {
    const uint32_t cur = P.parity, prev = cur ^ 1u;
    if (blockIdx.x * blockDim.x >= P.ctl->bound[prev]) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       at;
    const uint32_t total = queueOpen(&P.ctl->alive_count[prev][0], P.queue_capacity, i, at);
    if (0 == i) P.ctl->bound[cur] = total;
    const uint32_t rounded = (total + 255u) & ~255u;  // whole blocks take part in queuePush
    if (i >= rounded) return;
    const DScene&      s = c_scene;
    Counters           none{0, 0, 0};
    unsigned long long no_rays = 0;
    SectionTimer       tm;
    tm.start();
    bool     alive = false, defer = false;
    uint32_t slot = 0;
    if (i < total) {
        slot = P.alive_in[at];
        ScatterOut out;
        const HitRecord hr = loadHitRecord(P, slot);
        scatterVertex<false, SUBST, MULTI, false, false, false, true, WIDE>(P, s, slot, out, none, no_rays, tm, &hr);
        if (out.push) {
            const float4   o = P.pool.ray_o[slot], d = P.pool.ray_d[slot];
            const uint32_t meta = P.pool.q0.get(slot).y;
            const uint32_t depth = (meta >> 3) & 31u, flags = (meta >> 13) & 63u;
            Ray            ray = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
            Hit            hit;
#if ZYG_FAST
            sceneIntersectUniform<false, WIDE>(s, ray, depth, hit);
#else
            sceneIntersectUniform<true, false>(s, ray, depth, hit);
#endif
            if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
            const uint32_t ucount = walksUnoccluding(flags) ? unoccludingCandidates<false, false, true>(P, s, slot, vertexPlane<false>(), depth, ray, none) : 0u;
            shadeVertex<false, false, false, false>(P, slot, o, d, hit, ucount, alive, defer, none, tm);
            if (defer || alive) {  // wf_shade's second pass and the next tick's closure set-up read the hit from the pool
                P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
                P.pool.hit_prop[slot] = hit.prop;
                P.pool.ucount[slot] = ucount;
            }
        }
    }
    queuePush(P.q.heavy, &P.ctl->heavy_count[cur], P.queue_capacity, defer, slot);
    __syncthreads();  // queuePush's shared words are reused below
    queuePush(P.alive_out, &P.ctl->alive_count[cur][0], P.queue_capacity, alive, slot);
}

// ---------------------------------------------------------------- regenerating kernels
// Option "regenerate" (scenes walked through the prop tables, in-place NEE, single-sample lights): the pool is a few million slots,
// every one of which carries a path all the time: a slot whose path ends takes the batch's next camera sample in the same launch
// (wf_rscatter).  Thread i is slot i in both kernels: no extension / alive queues, no appends, no launches over thinning queues,
// and the records are plain arrays (Rec::stride = 16) that a wave reads and writes 64 slots = 1 KB at a time, where the tick loop's
// surviving slots thin out and every record costs its own line.  A slot is not its sample any more: Params::regen_sample names the
// sample (sampler seed, pixel, staging entry); the film is accumulated from the staging buffer in sample order as before, so the
// result is the tick loop's bit for bit.  A tick = wf_rscatter (scatter the living, start new paths in the free slots), wf_rtrace
// (closest hit + first shading pass for every slot), wf_shade's second pass for the few vertices with emission to evaluate (the
// heavy queue, one append per block).  The host asks every few ticks whether any slot is still alive.
// The free slots of 1024 consecutive ones are gathered in LDS, then the block's first waves start a new path in each: full waves in
// generatePath instead of every wave of wf_rscatter running it for its few free lanes, and one atomic per block on the batch's
// sample counter.  (A template so that only the unit that instantiates it emits it.)
constexpr uint32_t kRegenerateThreads = 1024;
template <bool POOL>
__global__ void __launch_bounds__(kRegenerateThreads) wf_rgenerate(Params P) {
    static_assert(!POOL, "the regenerating kernels are the plain instantiations");
    __shared__ uint32_t           free_slots[kRegenerateThreads];
    __shared__ uint32_t           wave_count[kRegenerateThreads / 64];
    __shared__ unsigned long long first_sample;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, lane = __lane_id(), wave = threadIdx.x >> 6;
    const bool     in = slot < P.regen_slots;
    const bool     is_free = in && kNull == P.regen_sample[slot];
    const unsigned long long fmask = __ballot(is_free);
    if (0 == lane) wave_count[wave] = uint32_t(__popcll(fmask));
    __syncthreads();
    uint32_t offset = uint32_t(__popcll(fmask & ((1ull << lane) - 1ull))), total = 0;
    for (uint32_t w = 0; w < kRegenerateThreads / 64; ++w) {
        const uint32_t n = wave_count[w];
        if (w < wave) offset += n;
        total += n;
    }
    if (is_free) free_slots[offset] = slot;
    if (0 == threadIdx.x) first_sample = total ? atomicAdd(&P.ctl->next_sample, (unsigned long long)total) : 0ull;
    __syncthreads();
    const unsigned long long mine = first_sample + threadIdx.x;
    bool                     started = false;
    if (threadIdx.x < total && mine < P.regen_samples) {
        const uint32_t target = free_slots[threadIdx.x];
        P.regen_sample[target] = uint32_t(mine);
        generatePath<false, false, false>(P, target, nullptr, nullptr, nullptr, uint32_t(mine));
        started = true;
    }
    if (P.regen_report) {  // the launch after which the host reads Control::regen_live
        const bool live = started || (in && !is_free);
        if (0 != __ballot(live) && 0 == lane) P.ctl->regen_live = 1u;
    }
}

template <bool WIDE>
__global__ void __launch_bounds__(256) ZYG_WAVES(ZYG_TRACE_SHADE_WAVES) wf_rtrace(Params P) {
    const uint32_t num_slots = P.regen_slots;
    const uint32_t cur = P.parity;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (0 == blockIdx.x && threadIdx.x < 64) queueReset(&P.ctl->heavy_count[cur ^ 1u]);  // the next tick's
    const DScene& scene = c_scene;
    Counters      none{0, 0, 0};
    SectionTimer  tm;
    tm.start();
    bool           alive = false, defer = false;
    const uint32_t sample = slot < num_slots ? P.regen_sample[slot] : kNull;
    if (kNull != sample) {
        PathHead fresh;
        loadHead(P, slot, fresh);
        const float4   thr = P.pool.thr[slot];
        const float4   o = P.pool.ray_o[slot], d = P.pool.ray_d[slot];
        const uint32_t depth = fresh.depth_surface, flags = fresh.flags & 63u;
        Ray            ray = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
        Hit            hit;
#if ZYG_FAST
        sceneIntersectUniform<false, WIDE>(scene, ray, depth, hit);
#else
        sceneIntersectUniform<true, false>(scene, ray, depth, hit);
#endif
        if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
        const uint32_t ucount = walksUnoccluding(flags) ? unoccludingCandidates<false, false, true>(P, scene, slot, vertexPlane<false>(), depth, ray, none) : 0u;
        shadeVertex<false, false, false, false, 2>(P, slot, o, d, hit, ucount, alive, defer, none, tm, &fresh, thr);
        if (defer || alive) {
            P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
            P.pool.hit_prop[slot] = hit.prop;
            P.pool.ucount[slot] = ucount;
        } else {
            P.regen_sample[slot] = kNull;  // the path ended: wf_rscatter starts a new one here
        }
    }
    queuePush(P.q.heavy, &P.ctl->heavy_count[cur], P.queue_capacity, defer, slot);
}

template <bool SUBST, bool WIDE>
__global__ void __launch_bounds__(256) ZYG_WAVES(SUBST ? ZYG_SCATTER_BIG_WAVES : ZYG_SCATTER_WAVES) wf_rscatter(Params P) {
    const uint32_t     num_slots = P.regen_slots;
    const uint32_t     slot = blockIdx.x * blockDim.x + threadIdx.x;
    const DScene&      s = c_scene;
    Counters           none{0, 0, 0};
    unsigned long long no_rays = 0;
    SectionTimer       tm;
    tm.start();
    bool live = false;
#if ZYG_EXPERIMENT_SKELETON  // timing experiment (wrong images): the path-state loads and stores of the plain-array layout alone
    if (slot < num_slots && kNull != P.regen_sample[slot]) {
        PathHead h;
        loadHead(P, slot, h);
        const float4   ro = P.pool.ray_o[slot], rd = P.pool.ray_d[slot];
        const uint4    hr = P.pool.hit[slot];
        const uint32_t hp = P.pool.hit_prop[slot];
        const float4   thr = P.pool.thr[slot];
        float4         ind = P.pool.indirect[slot];
        ind.x += thr.x * __uint_as_float(hr.x) + float(hp);
        P.pool.indirect[slot] = ind;
        P.pool.ray_o[slot] = ro;
        P.pool.ray_d[slot] = rd;
        P.pool.thr[slot] = thr;
        P.pool.vorigin[slot] = ro;
        P.pool.vgeo_n[slot] = rd;
        h.depth_surface += 1;
        storeHead(P, slot, h);
        live = true;
    }
#else
    if (slot < num_slots && kNull != P.regen_sample[slot]) {
        ScatterOut out;
        const HitRecord hr = loadHitRecord(P, slot);
        scatterVertex<false, SUBST, false, false, false, false, true, WIDE>(P, s, slot, out, none, no_rays, tm, &hr);
        live = out.push;
    }
#endif
    if (slot < num_slots && !live) P.regen_sample[slot] = kNull;  // the path ended (or there was none): wf_rgenerate fills the slot
}

// ---------------------------------------------------------------- wf_persistent
// The whole path in one kernel (option "persistent"; north_star: persistent threads, path state staged in LDS): a lane
// generates a camera sample, then loops trace -> shade -> scatter on it with the path's record block (the same 256 B + 32 B
// the wavefront kernels keep in HBM) in LDS, and takes the next sample of the batch when its path ends.  No queues, no state
// round trips through HBM; the price is one wave running every stage's code (the registers of the largest one, lanes idling
// through the rare heavy branches) and lanes of one wave sitting at different path depths.
// The film is still accumulated from the staging buffer in slot order, so the result is the wavefront kernels' bit for bit.
constexpr uint32_t kPersistentThreads = 128;
template <bool SUBST, bool MULTI, bool WIDE>
__global__ void __launch_bounds__(kPersistentThreads) ZYG_WAVES(2) wf_persistent(Params P, uint32_t num_paths) {
    __shared__ uint4 blocks[kPersistentThreads * (ZYG_POOL_STRIDE + ZYG_HIT_STRIDE) / 16];
    const DScene&    s = c_scene;
    Params           L = P;
    L.packed_prev = 0;  // (the lane's block in LDS has its second line at hand)
    char*            block = reinterpret_cast<char*>(blocks);
    char*            hits = block + kPersistentThreads * ZYG_POOL_STRIDE;
    const uint32_t   lane_block = threadIdx.x;
#define ZYG_LOCAL(rec, ptr) L.pool.rec.base = (ptr), L.pool.rec.mask = 0u, L.pool.rec.local = lane_block, L.pool.rec.stride = decltype(L.pool.rec)::kBlockStride
    ZYG_LOCAL(q0, block + kOffQ0);
    ZYG_LOCAL(q1, block + kOffQ1);
    ZYG_LOCAL(pcg, block + kOffPcg);
    ZYG_LOCAL(ray_o, block + kOffRayO);
    ZYG_LOCAL(ray_d, block + kOffRayD);
    ZYG_LOCAL(thr, block + kOffThr);
    ZYG_LOCAL(rec_p, block + kOffRecP);
    ZYG_LOCAL(rec_gn, block + kOffRecGn);
    ZYG_LOCAL(rec_n, block + kOffRecN);
    ZYG_LOCAL(rec_t, block + kOffRecT);
    ZYG_LOCAL(rec_b, block + kOffRecB);
    ZYG_LOCAL(vorigin, block + kOffVorigin);
    ZYG_LOCAL(vgeo_n, block + kOffVgeoN);
    ZYG_LOCAL(emission, block + kOffEmission);
    ZYG_LOCAL(direct, block + kOffDirect);
    ZYG_LOCAL(indirect, block + kOffIndirect);
    ZYG_LOCAL(hit, hits);
    ZYG_LOCAL(hit_prop, hits + 16);
    ZYG_LOCAL(ucount, hits + 20);
    ZYG_LOCAL(hit_sr, hits + 24);
#undef ZYG_LOCAL
    Counters           none{0, 0, 0};
    unsigned long long no_rays = 0;
    SectionTimer       tm;
    tm.start();
    bool     active = false, exhausted = false;
    uint32_t slot = 0;
    for (;;) {
        if (!active && !exhausted) {  // the next sample of the batch: one atomic per wave for the lanes that need one
            const unsigned long long need = __ballot(true);
            const uint32_t           count = uint32_t(__popcll(need)), rank = uint32_t(__popcll(need & ((1ull << __lane_id()) - 1ull)));
            unsigned long long       first = 0;
            if (0 == rank) first = atomicAdd(&P.ctl->next_sample, (unsigned long long)count);
            first = __shfl(first, __ffsll(need) - 1);
            const unsigned long long mine = first + rank;
            if (mine < num_paths) {
                slot = uint32_t(mine);
                generatePath<false, false>(L, slot);
                active = true;
            } else {
                exhausted = true;
            }
        }
        if (0 == __ballot(active)) break;
        if (active) {
            const float4   o = L.pool.ray_o[slot], d = L.pool.ray_d[slot];
            const uint32_t meta = L.pool.q0.get(slot).y;
            const uint32_t depth = (meta >> 3) & 31u, flags = (meta >> 13) & 63u;
            Ray            ray = Ray::init(xyz4(o), xyz4(d), 0.f, o.w);
            Hit            hit;
#if ZYG_FAST
            sceneIntersectUniform<false, WIDE>(s, ray, depth, hit);
#else
            sceneIntersectUniform<true, false>(s, ray, depth, hit);
#endif
            if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
            const uint32_t ucount = walksUnoccluding(flags) ? unoccludingCandidates<false, false, true>(L, s, slot, vertexPlane<false>(), depth, ray, none) : 0u;
            L.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
            L.pool.hit_prop[slot] = hit.prop;
            L.pool.ucount[slot] = ucount;
            bool alive = false, defer = false;
            shadeVertex<false, false, false, false>(L, slot, o, d, hit, ucount, alive, defer, none, tm);
            if (defer) {  // emission to evaluate: the complete connectLight (wf_shade's second pass)
                defer = false;
                shadeVertex<false, false, true, false>(L, slot, o, d, hit, ucount, alive, defer, none, tm);
            }
            if (alive) {
                ScatterOut out;
                const HitRecord hr = loadHitRecord(L, slot);
                scatterVertex<false, SUBST, MULTI, false, false, false, true, WIDE>(L, s, slot, out, none, no_rays, tm, &hr);
                alive = out.push;
            }
            active = alive;
        }
    }
}

}  // namespace zd
