// raster_fwd.hip -- forward rasteriser for gfx950 (MI355X).
//
// Replaces upstream kernels forward_face_index_map_{1,2} + forward_texture_sampling and the
// Python epilogue of RasterizeFunction.forward / rasterize_rgbad
// (/root/reference/meshreg/neurender/rasterize.py:87-103, 413-428).
//
// Design (not the upstream "every pixel loops over every face" scheme):
//   1a. face_records_kernel one thread per face: back-face cull + conservative pixel bbox (8 B) and, for live
//                          faces, a 96-B record {9 coordinates, pixel-space inverse, face index, vertex ids}
//                          at the face's own index.
//   1b. bin_boxes_kernel   ONE 1024-thread workgroup per image bins the image's boxes to screen tiles
//                          entirely in LDS (no global atomics, no memset): a counting pass (LDS integer
//                          atomics per bin), an exclusive scan over the bins and a fill pass produce per bin
//                          a contiguous list of 16-B records {bbox, face index}.  A bin
//                          is one 32x8 tile (a column of 2^s tiles for rasters beyond ~1400 pixels, so
//                          that the counters fit LDS).  Faces overlapping more than 8 bins go to a
//                          per-image "large" list instead, which every tile of the image scans.
//   2. raster_tile_kernel  one 256-thread workgroup per 32x8 screen tile (128-B output rows).  Tiles whose bin
//                          list and the image's large list are empty skip straight to the background
//                          fill.  Otherwise each of the 4 waves takes a quarter of (bin list + large
//                          list) -- independent 16-B loads per lane in flight --, ballots the records
//                          that touch the tile (all of them, for a one-tile bin) and compacts them (wave64
//                          ballot + popcount prefix) into a wave-private LDS ring.  The ring is consumed in
//                          batches of 32 faces, in three lock-step stages that keep the lanes
//                          full for the 5x5..16x16-pixel triangles of these meshes:
//                            S1 lane per face: load the 9 floats, invert the pixel-space
//                               matrix, park the face in an LDS face cache;
//                            S2 one lane per (face, bbox row) item -- the items of the batch are
//                               compacted with a prefix sum, 64 per pass --: the
//                               pixels a face covers on a row form ONE span (each edge test is
//                               monotone in x, also in floating point), found by three
//                               interleaved bisections with the exact predicate; the span's
//                               pixels are appended as fragments (face slot, x, y) to an LDS
//                               ring by ballot + popcount;
//                            S3 lane per fragment, 64 at a time: barycentrics (the 7 IEEE
//                               divisions), near/far test, depth test.
//                          Depth test = ds_min_u64 on a per-tile LDS z-buffer holding
//                          (ordered(zp) << 32 | face_index): a lexicographic min, i.e. exactly
//                          upstream's "strict < in ascending face order" (nearest face, lowest
//                          index on ties), independent of processing order.
//   3. resolve (same kernel) each thread owns one pixel: decode winner, recompute its
//                          barycentrics (bit-identical to the winning test), sample the
//                          texture, blend background, write every output plane once,
//                          already vertically flipped / NCHW for the image-space outputs.
// Vertex-colour mode (face_records_kernel<VC=true>, raster_tile_kernel<FUSED, VC=true>): geometry through the vertex
// indices, fill-back by index arithmetic, colours of the three vertices instead of a texture (bit-identical).
// The order of the records inside a bin list depends on the order of the LDS atomics; the image does not: the
// z-buffer key (depth, face index) makes the depth test a commutative minimum.
// HBM traffic: faces 36 B + 48 B + ~3 x 16 B records per live face, outputs written exactly once; no
// per-pixel memset, no sampling maps, no separate flip / permute / alpha / background pass.
#include <algorithm>

#include "mr_common.hpp"
#include "vertex_stage_device.hpp"

namespace mr {

constexpr int TILE_W = 32, TILE_H = 8;  // tile size in pixels (one pixel per thread)
constexpr int TPB = 256;              // threads per workgroup (4 waves)
constexpr int NB = 32;                // faces per batch (stage S1: one lane per face).  48 and 64 were measured (one
                                      // batch per wave for the ~145 faces of a typical geometry tile): the larger face
                                      // cache costs occupancy (5 / 4 instead of 7 waves per SIMD) and the kernel gets slower
constexpr int FC_STRIDE = 25;         // dwords per face-cache slot (odd: conflict-free ds_read_b32): v 9, inv 9, face index,
                                      // box, refined reciprocals of the three depths, "division-safe" flag
constexpr int FQCAP = 512;            // fragment ring capacity (>= 63 + 4 * 64, power of 2)

// Per-image header written by bin_faces_kernel.
struct ImageHdr {
    int n_live;   // live (front-facing, on-screen) faces = entries of the image's RecVerts array
    int n_large;  // faces overlapping more than SMALL_MAX_BINS bins: the image's large list
    int pad[6];
};
struct BinHdr {
    unsigned off, cnt;  // the bin's records: recs[image base + off .. + cnt)
};
// (TileList, the header of a launch's tile list, lives in mr_common.hpp: the warp kernels walk the same list)
// A tile's S1-S3 phase takes 6 us below 50 records and 16 - 20 us above 200 (workgroup timeline): handed out in screen
// order, the last heavy tiles start when the launch is nearly over and the chip drains for 20 us behind them.  The list
// therefore has two parts -- tiles with >= HEAVY_RECS records first -- so that the tail is made of light tiles.
constexpr unsigned HEAVY_RECS = 128;  // swept 32 .. 256 on the metric workload: 32-128 within 1 us, 160+ slower by 3-6 us

// {x0 | x1 << 16, y0 | y1 << 16, (virtual) face index, unused}
typedef uint4 FaceRec;
// Vertex-colour mode: the 9 coordinates of REAL face f0 (its own vertex order), dense array with a 48-B stride,
// so that the tile kernel fetches a face with ONE load phase instead of chasing record -> vertex indices ->
// vertices.  The reversed copy f0 + F0 is the same record read back to front.  (A 96-B record per live virtual
// face that also carried the pixel-space inverse and the vertex ids was measured in round 2: its scattered partial-line
// stores cost 12 us per launch and neither S1 nor the resolve step got faster -- they are latency-bound.  Round 6 measured
// it again with the set-up -- inverse + depth reciprocals of the live orientation -- written by the per-face pass inside the
// binning kernel, coalesced: the tile kernel's vector instructions fell from 27.8 M to 24.2 M per launch (S1 4.5 -> 2.4,
// resolve 6.3 -> 5.0) and its duration from 62.5 to 61.7 us, while the binning kernel, which IS issue-bound on the 128
// compute units it occupies, went from 28.5 to 37.2 us: profiles/r06_face_setup_records_experiment.txt.)
struct __attribute__((aligned(16))) RecVerts {
    float v[12];
};
static_assert(sizeof(RecVerts) == 48, "RecVerts is read as three float4");

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

constexpr int BIN_TPB = 1024;         // bin_boxes_kernel: one workgroup per image
constexpr int MAX_BINS = 8192;        // LDS counters (32 KB)
constexpr int SMALL_MAX_BINS = 8;     // a face overlapping more bins goes to the image's large list
constexpr int REC_CAP = SMALL_MAX_BINS + 1;  // record capacity per image, in units of F: 8 F binned + F large
constexpr int MAX_PARTS_DEV = 16;            // most workgroups per image of the binning pass (= MAX_PARTS of launch_bins)
constexpr int ARRIVE_STRIDE = 32;            // words between the arrival counters of two images

// PROLOGUE (pair steps): the parts of an image split its faces AT the hand / object boundary -- Kh parts share the Fh hand faces,
// K - Kh the Fo object faces, in proportion -- so that a part needs only the vertices of ITS side in LDS (hand faces index the
// hand's vertices, object faces the object's: warpbranch.py:49-55, the contract of the stacked mesh): one trip of the vertex
// stage per part instead of two at the metric workload.  Returns the part's real faces [r0, r0 + nr) and its side (0 hand, 1 object,
// -1: no split -- one part, or a mesh with one side only).
__host__ __device__ inline int pair_part_range(int part, int K, int Fh, int Fo, int& r0, int& nr) {
    const int F0 = Fh + Fo;
    if (K < 2 || Fh <= 0 || Fo <= 0) {
        r0 = (int)((int64_t)F0 * part / K); nr = (int)((int64_t)F0 * (part + 1) / K) - r0;
        return -1;
    }
    int Kh = (int)(((int64_t)K * Fh + F0 / 2) / F0);
    Kh = Kh < 1 ? 1 : (Kh > K - 1 ? K - 1 : Kh);
    if (part < Kh) {
        r0 = (int)((int64_t)Fh * part / Kh); nr = (int)((int64_t)Fh * (part + 1) / Kh) - r0;
        return 0;
    }
    const int q = part - Kh, Ko = K - Kh;
    r0 = (int)((int64_t)Fo * q / Ko); nr = (int)((int64_t)Fo * (q + 1) / Ko) - r0;
    r0 += Fh;
    return 1;
}

struct BinParams {
    const float* faces;      // !VC: [B,F,3,3]
    const float* verts;      // VC: [B,V,3]
    const int32_t* fidx;     // VC: [B,F0,3]
    ImageHdr* hdrs;          // [B]
    BinHdr* bins;            // [B, nbins]
    FaceRec* recs;           // [B, REC_CAP * F]
    RecVerts* rverts;        // VC: [B, F0] coordinates of the real faces (entries of dead faces stay unwritten)
    FaceBox* boxes;          // [B, F] pixel bbox per (virtual) face, empty = dead
    float* faces_inv;        // !VC, nullable: upstream's per-face inverse for the compatible API
    int V, F0, F, fill_back, is;
    int nbx, nby, ysh;       // bins per row / column; a bin is TILE_W x (TILE_H << ysh) pixels
    int lds_boxes;           // bin_boxes_kernel keeps the image's boxes in LDS between its two passes
    int dbg;                 // profiling experiments (scripts/fwd_vc_variants.py)
    // compacted list of the tiles with candidate records (sparse-tile launches; nullptr = none is built)
    TileList* tlist;         // counter, zeroed by face_records_kernel, bumped once per image by bin_boxes_kernel
    int64_t tile_cap;        // B * tiles: the list's heavy part starts at entry 0, its light part at entry tile_cap
    uint4* tile_ids;         // [2 * B * tiles] list entries {global tile id = image * tiles per image + tile, offset and
                             // count of the bin's records, length of the image's large list}, grouped by image: everything
                             // a tile's workgroup needs to start on its records after ONE scalar load
    uint8_t* tile_hit;       // [B, tiles, 4]: the binning pass writes the (zero) coverage bytes of the other tiles
    uint32_t* bg_ids;        // nullable (dense launches): [B * tiles] receives the global ids of the tiles WITHOUT candidates
    float* zero_fill;        // nullable: cleared by the binning pass (the matching backward's gradient buffer)
    int64_t zero_count;
    // round 5: `parts` workgroups per image, each binning a contiguous range of the image's faces (see bin_boxes_kernel)
    int B, parts;
    int box_cap;             // boxes the LDS copy holds (the largest part's; PROLOGUE: the projected vertices sit behind them)
    int split_fh, split_fo;  // > 0: the parts split the real faces at the hand / object boundary (pair_part_range; pair steps)
    int* part_cnt;           // [B, parts, nbins + 4]: a part's raw bin counters + {its large faces, its "everywhere" flag}
    unsigned* arrive;        // [B * ARRIVE_STRIDE] arrival counters of the images' parts, a 128-byte line each; ZERO on entry (per-face
                             // pass / the caller's clear) and again on exit (the last part to arrive re-zeroes its image's)
};

// Pass A, one thread per REAL face, grid = (ceil(F0 / 256), B): back-face cull + conservative pixel bbox of the
// face in its own orientation and -- vertex-colour mode with fill-back -- of its reversed copy (virtual face
// f0 + F0).  The order-independent part of face_box (pixel coordinates, bbox, sliver measure) is evaluated once
// for both.  Writes the box of every virtual face (8 B, empty = dead) and, vertex-colour mode, the gathered
// coordinates of the real face (dense, coalesced).  Nothing here depends on another face: the whole chip works on it.
template <bool VC>
__global__ void __launch_bounds__(256) face_records_kernel(BinParams p) {
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (p.tlist && b == 0 && f0 == 0) { p.tlist->n_heavy = 0u; p.tlist->n_light = 0u; p.tlist->n_bg = 0u; }  // (this launch precedes the binning pass)
    if (p.parts > 1 && f0 == 0) p.arrive[(int64_t)b * ARRIVE_STRIDE] = 0u;
    if (f0 >= p.F0) return;
    const int is = p.is;
    float f[9];
    if (!VC) {
        const float* g = p.faces + ((int64_t)b * p.F + f0) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = g[k];
        if (p.faces_inv && !backfacing(f)) {
            float inv[9];
            face_inverse(f, inv, is);
#pragma unroll
            for (int k = 0; k < 9; k++) p.faces_inv[((int64_t)b * p.F + f0) * 9 + k] = inv[k];
        }
    } else {
        const int32_t* ix = p.fidx + ((int64_t)b * p.F0 + f0) * 3;
        const int id[3] = {ix[0], ix[1], ix[2]};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float* g = p.verts + ((int64_t)b * p.V + id[k]) * 3;
            f[3 * k] = g[0]; f[3 * k + 1] = g[1]; f[3 * k + 2] = g[2];
        }
    }
    const bool two = VC && p.fill_back;
    BoxShared sh;
    face_box_shared(f, is, sh);
    const FaceBox b0 = face_box_orient<false>(f, is, sh);  // NaN / back-facing / off-screen -> empty
    FaceBox* box_b = p.boxes + (int64_t)b * p.F;
    box_b[f0] = b0;
    bool live = b0.x0 <= b0.x1;
    if (two) {
        const FaceBox b1 = face_box_orient<true>(f, is, sh);
        box_b[f0 + p.F0] = b1;
        live = live || b1.x0 <= b1.x1;
    }
    if (VC && live && !(p.dbg & 16)) {
        float4* rv = reinterpret_cast<float4*>(p.rverts + (int64_t)b * p.F0 + f0);
        rv[0] = make_float4(f[0], f[1], f[2], f[3]);
        rv[1] = make_float4(f[4], f[5], f[6], f[7]);
        // (second float: the face may take the shared-reciprocal division paths of the tile kernel, mr_common.hpp)
        rv[2] = make_float4(f[8], division_safe_face(f, is) ? 1.0f : 0.0f, 0.0f, 0.0f);
    }
}

// Pass B, grid = B, block = BIN_TPB: ONE workgroup bins the boxes of an image to screen tiles entirely in LDS (no
// global atomics, no memset): a counting pass (LDS integer atomics per bin), an exclusive scan over the bins and
// a fill pass.  Dynamic LDS: int cnt[nbins] | FaceBox boxes[F] (p.lds_boxes).
// (Measured and dropped, each SLOWER than the plain per-lane LDS atomics below -- the kernel is bound by the
// length of each wave's dependent instruction chain, not by LDS conflicts: combining the lanes that hit one bin
// with a ballot loop, folding runs of equal neighbours into one atomic, prefetching 8 iterations of boxes.)
#ifdef MR_WG_TIMELINE
__device__ unsigned long long mr_dbg_bin[1024 * 8];  // profiling builds: phase stamps of the binning pass per image
#define MR_BIN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) mr_dbg_bin[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define MR_BIN_STAMP(k) do { } while (0)
#endif

// Where its time goes (scripts/wg_timeline.py on a -DMR_WG_TIMELINE build, 7104 virtual faces, 1024 bins): 10 us inside the
// kernel per image -- counting pass 4.1 (of which the boxes' load round trip ~1.5), scan + headers 1.6, fill pass 3.9
// -- + ~3 us of launch and ~2 us between the first image's start and the last one's end.  Measured WITHOUT effect on
// those numbers, and not kept: walking the faces in a strided order so that a wave's lanes land in different bins,
// padding the counter rows against LDS bank conflicts (32 bins per row = 32 banks), four predicated straight-line atomics
// instead of the bin loops for faces within 2 x 2 bins.  What is left is memory latency at three dependent points
// (boxes in, headers / records out) around ~2 x 2 us of atomics.
// RECORDS (vertex-colour mode, boxes in LDS, tile list whose header the CALLER cleared -- MR_FLAG_TILE_LIST_CLEARED): the per-face
// pass runs inside this kernel -- a thread computes the boxes of its real faces (and their reversed copies) straight into the LDS
// copy and writes the faces' coordinates; the boxes never exist in global memory.  One launch and one dependent round trip (the
// boxes' read-back) less per render: 14 + 17.5 -> 21 us for the 2B = 128 meshes of a training pair.
// PARTS (round 5; round 6: no barrier).  One workgroup per image leaves half the chip idle at the 2B = 128 renders of a training
// pair and 15 of 16 compute units at config 3's 16 renders.  An image is therefore binned by `parts` workgroups, each taking a
// contiguous range of its faces through the per-face pass and the counting pass; a part then PUBLISHES its bin counters, its
// boxes and its scalars in global memory, counts itself on the image's arrival counter and LEAVES -- except the part that arrives
// last: it adds the parts' counters up, scans them, writes the bin headers and the image's tile-list entries and runs the fill
// pass over ALL boxes of the image.  Nobody waits for anybody: no assumption about which workgroups are resident together, about
// other kernels on the device or about the XCD a workgroup lands on (round 5's parts met at a spin barrier that needed all B x
// parts workgroups co-resident and trapped after ~2 s otherwise).  What crosses workgroups does so in agent-scope
// atomic stores / loads and one agent-scope atomic add per part (the "last block reduces" pattern; details at the exchange
// below).  The last arriver also re-zeroes the counter for the next launch.
// The order of the records inside a bin's list differs from the one-workgroup order -- as it already does from run to run (LDS
// atomics) -- and the image does not depend on it (z-buffer keys).
// PROLOGUE (round 6, pair steps: mr_render_flow_forward_pair): the vertex stage of the frame pair runs HERE instead of in a
// launch of its own in front (pair_prologue_kernel, 7.4 us for 2B = 128 meshes -- most of it a launch's floor).  The
// workgroup(s) of stack image b project the image's vertices into LDS (both frames' 2-D projections for the flow colour, the
// image's own frame through nr.projection: pair_vertex_of_frame -- vertex_stage_device.hpp, the arithmetic of
// flow_vertices_forward_body), the first part of a side writes that side's flow colours for the tile kernel, and every part
// converts its own range of the pair's int64 faces (written back as the image's int32 rows of the stacked faces, which the tile
// kernel's resolve reads).  The per-face pass then gathers its vertices from LDS instead of from global memory.  Nothing
// crosses workgroups: a part computes the vertices of ITS side of the mesh itself (pair_part_range: the parts split at the
// hand / object boundary; 778 or 1002 x ~250 instructions over 1024 threads), a single part all of them.
// PHASE (round 6, K > 1): the parts' exchange across a KERNEL boundary instead of through the memory side inside one launch.
// PHASE 1 = per-face pass + counting pass; the part leaves its counters, scalars and boxes in global memory with plain stores
// and is done.  PHASE 2 (a second launch of the same grid) = every part reads ALL parts' counters of its image, derives the
// merged counts, the scan and -- from the counts of the parts in front of it -- its OWN fill cursors per bin, and fills the records
// of its OWN faces; part 0 also writes what the image has once (bin headers, tile-list entries, image header).  The last-arriver
// form (PHASE 0 with K > 1) paid three dependent memory-side hops (~2 us each: publish -> arrival counter -> read back) and then
// ran merge, scan and the fill of the WHOLE image on one workgroup while the image's other parts had left; a launch boundary
// costs about as much as the hops, and behind it all parts work.  The order of a bin's records differs (by part, then by LDS
// atomic order) -- as it does from run to run -- and the image does not depend on it (z-buffer keys).
template <bool RECORDS, bool PROLOGUE, int PHASE = 0>
__device__ __forceinline__ void bin_boxes_body(const BinParams& p, const PairPrologue& pro) {
    static_assert(RECORDS || !PROLOGUE, "the prologue feeds the per-face pass");
    static_assert(PHASE == 0 || !PROLOGUE || PHASE == 1, "the fill launch has no vertex stage");
    MR_BIN_STAMP(0);
    extern __shared__ int bin_smem[];
    __shared__ int s_large, s_nlarge, s_lbase, s_hbase, s_bbase, s_everywhere, s_last;
    // listed launches: bit 30 of a bin's counter = "a face of the image's large list overlaps this bin" (counts stay far below)
    constexpr int LARGE_BIT = 1 << 30, CNT_MASK = LARGE_BIT - 1;
    __shared__ unsigned long long wsum[BIN_TPB / MR_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.parts;
    if (PHASE != 2 && p.zero_fill)
        for (int64_t i = (int64_t)blockIdx.x * BIN_TPB + tid; i < p.zero_count; i += (int64_t)gridDim.x * BIN_TPB) p.zero_fill[i] = 0.0f;
    int b = blockIdx.x, part = 0;
    if (K > 1) {  // workgroup -> (image, part): the parts of image b on XCD b % 8
        const unsigned q = blockIdx.x >> 3;
        part = (int)(q % (unsigned)K);
        b = (int)(q / (unsigned)K) * 8 + (int)(blockIdx.x & 7u);
        if (b >= p.B) return;
    }
    const int nbins = p.nbx * p.nby, nbins4 = (nbins + 3) & ~3;
    int* cnt = bin_smem;
    FaceBox* sbox = reinterpret_cast<FaceBox*>(bin_smem + nbins4);
    // PROLOGUE: the image's projected vertices [V][3], behind the boxes of the largest part (launch_bins sizes both)
    float* sverts = reinterpret_cast<float*>(sbox + p.box_cap);
    FaceBox* box_b = p.boxes + (int64_t)b * p.F;
    // this part's faces: real faces [r0, r0 + nr) and (RECORDS with fill-back) their reversed copies F0 + [r0, r0 + nr), or
    // (!RECORDS) virtual faces [r0, r0 + nr); local index j < nv -> face fn_of(j)
    const int nsplit = RECORDS ? p.F0 : p.F;
    int r0 = (int)((int64_t)nsplit * part / K), nr = (int)((int64_t)nsplit * (part + 1) / K) - r0;
    int side = -1;  // pair steps: the part's faces are all hand faces (0) / all object faces (1): pair_part_range
    if (RECORDS && p.split_fh > 0) side = pair_part_range(part, K, p.split_fh, p.split_fo, r0, nr);
    const bool two = RECORDS && p.fill_back != 0;
    const int nv = two ? 2 * nr : nr;
    auto fn_of = [&](int j) { return j < nr ? r0 + j : p.F0 + r0 + (j - nr); };

    for (int i = tid; i < nbins; i += BIN_TPB) cnt[i] = 0;
    if (tid == 0) { s_large = 0; s_nlarge = 0; s_everywhere = 0; s_last = 1; }
    __syncthreads();
    MR_BIN_STAMP(1);

    if constexpr (RECORDS && PHASE != 2) {
        // the per-face pass (face_records_kernel<true>'s arithmetic): indices of REC_PF faces per thread requested together,
        // then their vertices, then the boxes of both orientations into the LDS copy
        constexpr int REC_PF = 4;
        int id[REC_PF][3];
        // (PROLOGUE: stack image b is frame b / (B / 2) of pair b % (B / 2); its faces come from the pair's int64 tensors)
        const int pairs = p.B >> 1, pb = PROLOGUE ? b % max(pairs, 1) : 0, frame = PROLOGUE ? b / max(pairs, 1) : 0;
        auto load_ids = [&](int base) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < REC_PF; k++) {
                // (a slot none of the workgroup's threads has a face for is skipped -- uniform: a part of 2000 faces uses two of
                // the four, one of 222 a single one; their clamped duplicate requests cost the count launch 2 us at config 3)
                if (base + k * BIN_TPB >= nr) continue;
                const int fr = r0 + min(base + k * BIN_TPB + tid, nr - 1);
                if constexpr (PROLOGUE) {
#pragma unroll
                    for (int v = 0; v < 3; v++) id[k][v] = pair_face_index(pro.f, pb, fr, v);
                } else {
                    const int32_t* ix = p.fidx + ((int64_t)b * p.F0 + fr) * 3;
                    id[k][0] = ix[0]; id[k][1] = ix[1]; id[k][2] = ix[2];
                }
            }
        };
        if constexpr (PROLOGUE) {
            if (nr > 0) load_ids(0);  // (in flight beside the vertex stage below)
            PairCamera cam;
            load_pair_camera(pro.v, pb, cam);
            float* cols_b = (frame == 0 ? pro.v.cols12 : pro.v.cols21) + (int64_t)pb * p.V * 3;
            // the vertices of the part's side only (all of them without a split); the FIRST part of a side writes its colours
            const int v_lo = side == 1 ? pro.v.split : 0, v_hi = side == 0 ? pro.v.split : p.V;
            const bool writes = side < 0 ? part == 0 : (r0 == 0 || r0 == pro.f.Fh);
            for (int vi = v_lo + tid; vi < v_hi; vi += BIN_TPB) {
                float n[3], c[2];
                pair_vertex_of_frame(pro.v, cam, pb, vi, frame, n, c);
                sverts[vi * 3] = n[0]; sverts[vi * 3 + 1] = n[1]; sverts[vi * 3 + 2] = n[2];
                if (writes) { cols_b[vi * 3] = c[0]; cols_b[vi * 3 + 1] = c[1]; cols_b[vi * 3 + 2] = 1.0f; }
            }
            __syncthreads();
        }
        for (int base = 0; base < nr; base += REC_PF * BIN_TPB) {
            if (!PROLOGUE || base > 0) load_ids(base);
            if constexpr (PROLOGUE) {
                int32_t* f2 = pro.f.out + ((int64_t)b * p.F0 + r0) * 3;
#pragma unroll
                for (int k = 0; k < REC_PF; k++) {
                    const int j = base + k * BIN_TPB + tid;
                    if (j < nr) { f2[j * 3] = id[k][0]; f2[j * 3 + 1] = id[k][1]; f2[j * 3 + 2] = id[k][2]; }
                }
            }
            float f[REC_PF][9];
#pragma unroll
            for (int k = 0; k < REC_PF; k++)
#pragma unroll
                for (int v = 0; v < 3; v++) {
                    if (base + k * BIN_TPB >= nr) continue;  // (uniform: see load_ids)
                    if constexpr (PROLOGUE) {
                        const float* g = sverts + id[k][v] * 3;
                        f[k][3 * v] = g[0]; f[k][3 * v + 1] = g[1]; f[k][3 * v + 2] = g[2];
                    } else {
                        const float* g = p.verts + ((int64_t)b * p.V + id[k][v]) * 3;
                        f[k][3 * v] = g[0]; f[k][3 * v + 1] = g[1]; f[k][3 * v + 2] = g[2];
                    }
                }
#pragma unroll
            for (int k = 0; k < REC_PF; k++) {
                const int j = base + k * BIN_TPB + tid;
                if (j >= nr) continue;
                BoxShared sh;
                face_box_shared(f[k], p.is, sh);
                const FaceBox b0 = face_box_orient<false>(f[k], p.is, sh);
                sbox[j] = b0;
                bool live = b0.x0 <= b0.x1;
                if (two) {
                    const FaceBox b1 = face_box_orient<true>(f[k], p.is, sh);
                    sbox[nr + j] = b1;
                    live = live || b1.x0 <= b1.x1;
                }
                if (live) {
                    float4* rv = reinterpret_cast<float4*>(p.rverts + (int64_t)b * p.F0 + r0 + j);
                    rv[0] = make_float4(f[k][0], f[k][1], f[k][2], f[k][3]);
                    rv[1] = make_float4(f[k][4], f[k][5], f[k][6], f[k][7]);
                    rv[2] = make_float4(f[k][8], division_safe_face(f[k], p.is) ? 1.0f : 0.0f, 0.0f, 0.0f);
                }
            }
        }
        __syncthreads();
    }
    // pass 1: records per bin.  The boxes of BIN_PF trips are requested together (unconditional loads from clamped
    // addresses): one load round trip per BIN_PF x 1024 faces instead of one per trip -- a hand + object mesh (7104
    // virtual faces) is seven trips, i.e. seven dependent round trips of the single workgroup an image has.
    constexpr int BIN_PF = 8;
    for (int base = 0; base < nv && PHASE != 2; base += BIN_PF * BIN_TPB) {
        FaceBox bxs[BIN_PF];
#pragma unroll
        for (int k = 0; k < BIN_PF; k++) {
            const int jc = min(base + k * BIN_TPB + tid, nv - 1);
            bxs[k] = RECORDS ? sbox[jc] : box_b[r0 + jc];
        }
#pragma unroll
        for (int k = 0; k < BIN_PF; k++) {
            const int j = base + k * BIN_TPB + tid;
            const FaceBox bx = bxs[k];
            if (!RECORDS && j < nv && p.lds_boxes) sbox[j] = bx;
            if (j >= nv || bx.x0 > bx.x1) continue;
            const int bx0 = bx.x0 / TILE_W, bx1 = bx.x1 / TILE_W;
            const int by0 = bx.y0 >> (3 + p.ysh), by1 = bx.y1 >> (3 + p.ysh);
            if ((bx1 - bx0 + 1) * (by1 - by0 + 1) > SMALL_MAX_BINS) {  // large list (filled in pass 2)
                atomicAdd(&s_nlarge, 1);
                if (p.tlist) {
                    // ... which only the tiles under the face's box have to walk (round 4; until then one large face made
                    // every tile of its image a listed tile that walks the whole large list -- at 640 x 640, where the 28
                    // wrist-closing faces of a hand span more than 8 bins, 86 000 listed tiles instead of 19 000)
                    if ((bx1 - bx0 + 1) * (by1 - by0 + 1) * 4 > nbins) s_everywhere = 1;  // (degenerate faces: full-screen boxes)
                    else
                        for (int y = by0; y <= by1; y++)
                            for (int x = bx0; x <= bx1; x++) atomicOr(&cnt[y * p.nbx + x], LARGE_BIT);
                }
                continue;
            }
            for (int y = by0; y <= by1; y++)
                for (int x = bx0; x <= bx1; x++) atomicAdd(&cnt[y * p.nbx + x], 1);
        }
    }
    __syncthreads();
    MR_BIN_STAMP(2);

    const int per = (nbins + BIN_TPB - 1) / BIN_TPB;
    const int i0 = min(tid * per, nbins), i1 = min(i0 + per, nbins);
    int before[MAX_BINS / BIN_TPB];  // PHASE 2: records the parts IN FRONT of this one put into the thread's bins
#pragma unroll
    for (int u = 0; u < MAX_BINS / BIN_TPB; u++) before[u] = 0;
    if (K > 1 && PHASE == 1) {
        // the count launch: this part's raw counters, scalars and (RECORDS: they exist in LDS only) boxes to global memory,
        // plain stores -- the fill launch behind the kernel boundary reads them
        const int pstride = (nbins + 4 + 31) & ~31;
        int* pub = p.part_cnt + ((int64_t)b * K + part) * pstride;
        for (int i = tid; i < nbins; i += BIN_TPB) pub[i] = cnt[i];
        if (tid == 0) { pub[nbins] = s_nlarge; pub[nbins + 1] = s_everywhere; }
        if (RECORDS)
            for (int j = tid; j < nv; j += BIN_TPB) box_b[fn_of(j)] = sbox[j];
        return;
    }
    if (K > 1 && PHASE == 2) {
        // the fill launch: all parts' counters of the thread's bins (plain loads: written before the kernel boundary) -> merged
        // counts for the scan, and the share of the parts in front of this one = where ITS records start inside a bin's list
        const int pstride = (nbins + 4 + 31) & ~31;
        const int* all = p.part_cnt + (int64_t)b * K * pstride;
        for (int i = i0; i < i1; i++) {
            int raw[MAX_PARTS_DEV];
#pragma unroll
            for (int k = 0; k < MAX_PARTS_DEV; k++) raw[k] = k < K ? all[(int64_t)k * pstride + i] : 0;
            int tot = 0, lg = 0, bef = 0;
#pragma unroll
            for (int k = 0; k < MAX_PARTS_DEV; k++) {
                tot += raw[k] & CNT_MASK;
                lg |= raw[k] & LARGE_BIT;
                bef += k < part ? (raw[k] & CNT_MASK) : 0;
            }
            cnt[i] = tot | lg;
#pragma unroll
            for (int u = 0; u < MAX_BINS / BIN_TPB; u++)
                if (u == i - i0) before[u] = bef;
        }
        if (tid < MR_WAVE) {  // (wave 0; lane k < K: part k's scalars)
            int nl = 0, e = 0, nlb = 0;
            if (tid < K) {
                nl = all[(int64_t)tid * pstride + nbins];
                e = all[(int64_t)tid * pstride + nbins + 1];
                nlb = tid < part ? nl : 0;
            }
#pragma unroll
            for (int off = 1; off < MAX_PARTS_DEV; off <<= 1) {
                nl += __shfl_xor(nl, off); e |= __shfl_xor(e, off); nlb += __shfl_xor(nlb, off);
            }
            if (tid == 0) { s_nlarge = nl; s_everywhere = e; s_large = nlb; }  // (its large faces go behind the earlier parts')
        }
        __syncthreads();
    }
    if (K > 1 && PHASE == 0) {
        // publish this part's counters, scalars and (RECORDS: they exist in LDS only) boxes; count it on the image's arrival
        // counter; every part but the last one to arrive is done.  Everything that crosses workgroups here goes through
        // AGENT-scope atomic stores / loads (sc1: written through to / read from the memory side, past the XCD's L2) and the
        // arrival counter through an agent-scope atomic add: coherent whichever XCDs the parts run on, and without the L2
        // write-back of a release FENCE (measured, the formal form -- plain stores, agent-scope release fence + acquire fence
        // around the add -- took this kernel from 26.8 to 36.5 us in the step at 2B = 128: buffer_wbl2 flushes the ~2 MB of
        // vertex records the XCD's workgroups have just written, which nobody in this kernel reads).  Order: a part's stores
        // have completed (s_waitcnt vmcnt(0)) before its barrier, the add is issued behind the barrier.
        const int pstride = (nbins + 4 + 31) & ~31;  // (a part's counters start on a 128-byte line of their own)
        int* pub = p.part_cnt + ((int64_t)b * K + part) * pstride;
        for (int i = tid; i < nbins; i += BIN_TPB) __hip_atomic_store(&pub[i], cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&pub[nbins], s_nlarge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&pub[nbins + 1], s_everywhere, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        static_assert(sizeof(FaceBox) == 8, "boxes cross workgroups as 64-bit atomic words");
        unsigned long long* box_w = reinterpret_cast<unsigned long long*>(box_b);
        if (RECORDS)
            for (int j = tid; j < nv; j += BIN_TPB)
                __hip_atomic_store(&box_w[fn_of(j)], reinterpret_cast<const unsigned long long*>(sbox)[j], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MR_BIN_STAMP(6);
        if (tid == 0) {
            unsigned* arr = p.arrive + (int64_t)b * ARRIVE_STRIDE;
            const unsigned old = __hip_atomic_fetch_add(arr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == (unsigned)K - 1u) ? 1 : 0;
            if (s_last) __hip_atomic_store(arr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the next launch counts from zero)
        }
        __syncthreads();
        if (!s_last) return;
        MR_BIN_STAMP(7);
        const int* all = p.part_cnt + (int64_t)b * K * pstride;
        for (int i = i0; i < i1; i++) {
            int raw[MAX_PARTS_DEV];  // (all parts' counters of the bin in flight together)
#pragma unroll
            for (int k = 0; k < MAX_PARTS_DEV; k++)
                raw[k] = k < K ? __hip_atomic_load(all + (int64_t)k * pstride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            int tot = 0, lg = 0;
#pragma unroll
            for (int k = 0; k < MAX_PARTS_DEV; k++) {
                tot += raw[k] & CNT_MASK;
                lg |= raw[k] & LARGE_BIT;
            }
            cnt[i] = tot | lg;
        }
        if (tid < MR_WAVE) {  // (wave 0; lane k < K: part k's scalars)
            int nl = 0, e = 0;
            if (tid < K) {
                nl = __hip_atomic_load(all + (int64_t)tid * pstride + nbins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                e = __hip_atomic_load(all + (int64_t)tid * pstride + nbins + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int off = 1; off < MAX_PARTS_DEV; off <<= 1) {
                nl += __shfl_xor(nl, off); e |= __shfl_xor(e, off);
            }
            if (tid == 0) { s_nlarge = nl; s_everywhere = e; }
        }
        __syncthreads();
    }
    // (whoever gets here writes what the image has once -- bin headers, tile-list entries, image header; of the fill launch's parts: part 0)
    const bool lead = PHASE != 2 || K == 1 || part == 0;

    // exclusive scan of the bin counts (each thread owns a run of consecutive bins); headers out, counters
    // become fill cursors.  The high half of the scanned value counts the bins that hold candidates (every bin,
    // if the image has a large list): the places of the image's entries in the launch's tile list.
    const bool all_large = p.tlist && s_nlarge > 0 && s_everywhere != 0;
    unsigned long long local = 0;
    for (int i = i0; i < i1; i++) {
        const int raw = cnt[i], c = raw & CNT_MASK;
        const bool lg = all_large || (raw & LARGE_BIT);  // this bin's tile walks the large list
        // (bits 32-47: bins that hold candidates, bits 48-63: those of them that are heavy; at most 8192 bins per image)
        const unsigned nrec_i = (unsigned)c + (lg ? (unsigned)s_nlarge : 0u);
        local += (unsigned long long)(unsigned)c | ((unsigned long long)((c > 0 || lg) ? 1u : 0u) << 32) |
                 ((unsigned long long)((p.tlist && nrec_i >= HEAVY_RECS) ? 1u : 0u) << 48);
    }
    unsigned long long incl = local;
#pragma unroll
    for (int off = 1; off < MR_WAVE; off <<= 1) {
        const unsigned long long up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == MR_WAVE - 1) wsum[wave] = incl;
    __syncthreads();
    unsigned long long base64 = incl - local, total64 = 0;
    for (int w = 0; w < BIN_TPB / MR_WAVE; w++) {
        const unsigned long long ws = wsum[w];
        if (w < wave) base64 += ws;
        total64 += ws;
    }
    int base = (int)(unsigned)(base64 & 0xffffffffull);
    // The image's places in the launch's tile list: three device-scope atomics whose results are needed only when the entries are
    // written -- BEHIND the fill pass (round 6: the ~2 us memory-side round trip used to sit between the scan and the fill, with
    // the whole workgroup waiting at the barrier behind it; scripts/pair_bin_timeline.py).  The results stay in thread 0's
    // registers until then.
    int got_h = 0, got_l = 0, got_b = 0;
    if (p.tlist && lead && tid == 0) {
        const unsigned n_ne = (unsigned)(total64 >> 32) & 0xffffu, n_hv = (unsigned)(total64 >> 48);
        got_h = (int)atomicAdd(&p.tlist->n_heavy, n_hv);
        got_l = (int)atomicAdd(&p.tlist->n_light, n_ne - n_hv);
        if (p.bg_ids) got_b = (int)atomicAdd(&p.tlist->n_bg, (unsigned)nbins - n_ne);
    }
    BinHdr* bh = p.bins + (int64_t)b * nbins;
    unsigned livebits = 0u, lgbits = 0u;  // (per <= MAX_BINS / BIN_TPB = 8 bins per thread)
    for (int i = i0; i < i1; i++) {
        const int raw = cnt[i], c = raw & CNT_MASK;
        const bool lg = all_large || (raw & LARGE_BIT);
        if (lead) {
            BinHdr h;
            h.off = (unsigned)base; h.cnt = (unsigned)c;
            bh[i] = h;
        }
        int bef = 0;  // (selected, not indexed: the array stays in registers)
#pragma unroll
        for (int u = 0; u < MAX_BINS / BIN_TPB; u++) bef = (PHASE == 2 && u == i - i0) ? before[u] : bef;
        cnt[i] = base + bef;  // (fill cursor; part 0's -- the tile list's writer -- is the bin's offset)
        base += c;
        livebits |= ((c > 0 || lg) ? 1u : 0u) << (i - i0);
        lgbits |= (lg ? 1u : 0u) << (i - i0);
    }
    __syncthreads();
    MR_BIN_STAMP(3);
    auto write_tile_list = [&]() __attribute__((always_inline)) {
        if (!(p.tlist && lead)) return;  // (uniform)
        // the image's tiles with candidates go to the launch's tile list (a bin IS a tile here: ysh == 0), the
        // others get their (zero) coverage bytes now -- no workgroup is dispatched for them
        if (tid == 0) { s_hbase = got_h; s_lbase = got_l; s_bbase = got_b; }
        __syncthreads();
        const unsigned hv_before = (unsigned)(base64 >> 48), ne_before = (unsigned)(base64 >> 32) & 0xffffu;
        unsigned at_h = (unsigned)s_hbase + hv_before;                                   // heavy part: from the front
        unsigned at_l = (unsigned)p.tile_cap + (unsigned)s_lbase + (ne_before - hv_before);  // light part: second half
        unsigned at_b = p.bg_ids ? (unsigned)s_bbase + ((unsigned)i0 - ne_before) : 0u;     // tiles without candidates
        uint32_t* hit32 = reinterpret_cast<uint32_t*>(p.tile_hit) + (int64_t)b * nbins;
        for (int i = i0; i < i1; i++) {
            const bool live = (livebits >> (i - i0)) & 1u;
            const unsigned nlarge = ((lgbits >> (i - i0)) & 1u) ? (unsigned)s_nlarge : 0u;
            // (the bin's offset and count: the header this thread wrote above -- the LDS cursors have moved on by now)
            const BinHdr h = bh[i];
            const uint4 ent = make_uint4((unsigned)(b * nbins + i), h.off, h.cnt, nlarge);
            if (live && h.cnt + nlarge >= HEAVY_RECS) p.tile_ids[at_h++] = ent;
            else if (live) p.tile_ids[at_l++] = ent;
            else if (p.bg_ids) p.bg_ids[at_b++] = (unsigned)(b * nbins + i);
            else hit32[i] = 0u;
        }
    };
    MR_BIN_STAMP(4);
    if (p.dbg & 2) {  // (profiling: the binning pass without its fill)
        write_tile_list();
        return;
    }

    // pass 2: fill
    FaceRec* recs_b = p.recs + (int64_t)b * REC_CAP * p.F;
    FaceRec* large_b = recs_b + (int64_t)SMALL_MAX_BINS * p.F;
    auto fill_face = [&](const int fn, const FaceBox bx) __attribute__((always_inline)) {
        if (bx.x0 > bx.x1) return;
        FaceRec rec;
        rec.x = (unsigned)(unsigned short)bx.x0 | ((unsigned)(unsigned short)bx.x1 << 16);
        rec.y = (unsigned)(unsigned short)bx.y0 | ((unsigned)(unsigned short)bx.y1 << 16);
        rec.z = (unsigned)fn;
        rec.w = 0u;
        const int bx0 = bx.x0 / TILE_W, bx1 = bx.x1 / TILE_W;
        const int by0 = bx.y0 >> (3 + p.ysh), by1 = bx.y1 >> (3 + p.ysh);
        if ((bx1 - bx0 + 1) * (by1 - by0 + 1) > SMALL_MAX_BINS) {
            large_b[atomicAdd(&s_large, 1)] = rec;
        } else {
            for (int y = by0; y <= by1; y++)
                for (int x = bx0; x <= bx1; x++) recs_b[atomicAdd(&cnt[y * p.nbx + x], 1)] = rec;
        }
    };
    // (two loops, not one over `lds_boxes ? sbox[j] : box_b[fn]`: the merged pointer is a generic one, and every box
    // then costs two dependent FLAT loads with a full wait each, also when it sits in LDS)
    if (K > 1 && PHASE == 2) {
        // the fill launch: every part fills the records of ITS faces, from the boxes its count launch left (FILL_PF trips together)
        constexpr int FILL_PF = 4;
        for (int base = 0; base < nv; base += FILL_PF * BIN_TPB) {
            FaceBox bxs[FILL_PF];
#pragma unroll
            for (int k = 0; k < FILL_PF; k++) bxs[k] = box_b[fn_of(min(base + k * BIN_TPB + tid, nv - 1))];
#pragma unroll
            for (int k = 0; k < FILL_PF; k++) {
                const int j = base + k * BIN_TPB + tid;
                if (j < nv) fill_face(fn_of(j), bxs[k]);
            }
        }
    } else if (K > 1) {
        // the last arriver fills for the whole image, from the boxes all parts left in global memory (FILL_PF trips requested
        // together: a hand + object mesh's 7104 boxes in ONE memory-side round trip -- two of ~2 us each with four per thread,
        // scripts/pair_bin_timeline.py)
        constexpr int FILL_PF = 8;
        for (int base = 0; base < p.F; base += FILL_PF * BIN_TPB) {
            FaceBox bxs[FILL_PF];
#pragma unroll
            for (int k = 0; k < FILL_PF; k++) {
                const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(box_b) + min(base + k * BIN_TPB + tid, p.F - 1),
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bxs[k] = *reinterpret_cast<const FaceBox*>(&w);
            }
#pragma unroll
            for (int k = 0; k < FILL_PF; k++) {
                const int fn = base + k * BIN_TPB + tid;
                if (fn < p.F) fill_face(fn, bxs[k]);
            }
        }
    } else if (p.lds_boxes) {
        for (int j = tid; j < nv; j += BIN_TPB) fill_face(fn_of(j), sbox[j]);
    } else {
        for (int j = tid; j < nv; j += BIN_TPB) fill_face(fn_of(j), box_b[fn_of(j)]);
    }
    MR_BIN_STAMP(5);
    write_tile_list();
    if (lead && tid == 0) {
        ImageHdr h;
        h.n_live = 0; h.n_large = s_nlarge;
#pragma unroll
        for (int k = 0; k < 6; k++) h.pad[k] = 0;
        p.hdrs[b] = h;
    }
}

template <bool RECORDS>
__global__ void __launch_bounds__(BIN_TPB) bin_boxes_kernel(BinParams p) {
    bin_boxes_body<RECORDS, false>(p, PairPrologue{});
}
__global__ void __launch_bounds__(BIN_TPB) bin_boxes_prologue_kernel(BinParams p, PairPrologue pro) {
    bin_boxes_body<true, true>(p, pro);
}
// the parts' two launches (PHASE): count (with or without the pair's vertex stage in front), fill
template <bool RECORDS>
__global__ void __launch_bounds__(BIN_TPB) bin_count_kernel(BinParams p) {
    bin_boxes_body<RECORDS, false, 1>(p, PairPrologue{});
}
__global__ void __launch_bounds__(BIN_TPB) bin_count_prologue_kernel(BinParams p, PairPrologue pro) {
    bin_boxes_body<true, true, 1>(p, pro);
}
template <bool RECORDS>
__global__ void __launch_bounds__(BIN_TPB) bin_fill_kernel(BinParams p) {
    bin_boxes_body<RECORDS, false, 2>(p, PairPrologue{});
}

struct FwdParams {
    const float* faces;
    const ImageHdr* hdrs;
    const BinHdr* bins;
    const FaceRec* recs;
    const RecVerts* rverts;
    int nbx, nby, ysh;
    const float* textures;
    const float* background;
    int bg_stride;
    float* rgb;            // FUSED: [B,3,is,is] image orientation
    float* alpha;          // FUSED: [B,is,is] image orientation
    float* depth;          // FUSED: image orientation; COMPAT: raster orientation
    int32_t* fim;          // raster orientation
    float* weight;         // [B,is,is,3] raster orientation (nullable in flow mode)
    // flow mode (mr_render_flow_forward): the training path's output set
    float* mask;           // FUSED: [B,is,is] image orientation: (alpha > thresh) * keep_lut[face + 1]; nullable
    const float* keep_lut; // nullable: 0 for ignored faces, 1 otherwise, indexed by face index + 1
    int n_lut;
    float alpha_thresh;
    int rgb_channels;      // 3, or 2: the third colour plane is left untouched
    float4* rec4;          // pair steps (round 6), nullable: [B,is,is] image orientation, ONE 16-byte record per pixel {colour 0,
                           // colour 1, alpha, mask} INSTEAD of the rgb / alpha / mask planes -- the fused warp forward then asks for
                           // a pixel's four values in one load instead of four (it is bound by its count of memory requests)
    int32_t* vid_map;      // VC flow mode, nullable: [B,is,is,3] raster orientation: the winner's vertex ids; with it
                           // `weight` receives the three SAMPLING weights of the colour taps instead of the
                           // barycentrics -- all the colour backward needs, in one load round trip per pixel
    uint8_t* tile_hit;     // nullable: [B, tiles, 4] 1 = wave w (rows 2w, 2w + 1) of the tile covers a pixel
    int sparse_tiles;      // tiles without candidate faces write their coverage bytes (0) and NOTHING else
    int sparse_wd;         // weight / depth are written at covered pixels only (their only reader, the colour
                           // backward, looks at nothing else)
    float* face_inv_map;   // [B,is,is,9] raster orientation (nullable)
    int B, F, is, ts;
    float near_, far_, eps;
    int tiles_x, tiles_y;  // tiles per row / per column
    const unsigned long long* keys;  // validation only: precomputed z-buffer keys (skip the scan)
    int dbg;                         // profiling experiments (flags >> 8)
    // listed launches (sparse tiles): the binning pass's list of tiles with candidates
    const TileList* tlist;
    const uint4* tile_ids;           // heavy entries [0, n_heavy), light entries [tile_cap, tile_cap + n_light)
    unsigned tile_cap;
    const uint32_t* bg_ids;          // dense listed launches: ids of the tlist->n_bg tiles without candidates, which the
                                     // workgroups of the launch stream as background between them (grid stride)
    uint32_t* tile_count_out;        // nullable, any device-writable address (e.g. pinned host memory): the list
                                     // length of this launch, for the caller's next grid-size guess
    // vertex-colour mode (VC): indexed geometry + per-vertex colours, fill-back done by index
    // arithmetic: virtual face fn >= F0 is face fn - F0 with its vertex order reversed
    const float* verts;              // [B,V,3] projected vertices (x,y NDC, z metric)
    const int32_t* fidx;             // [B,F0,3] vertex indices
    const float* vcolors;            // [B,V,3]
    int V, F0;
    int texel;                       // texel layout code of the vertex-colour texture (mr_common.hpp: texel_vertex)
};

// the 9 coordinates of (virtual) face fn in ONE load phase: from the faces tensor, or (VC) from the gathered
// coordinates of real face fn mod F0, read back to front for the reversed copy
template <bool VC>
__device__ __forceinline__ bool load_face_coords(const FwdParams& p, const RecVerts* rv_b, int b, int fn, float* v) {
    if (!VC) {
        const float* g = p.faces + ((int64_t)b * p.F + fn) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = g[k];
        return false;  // (the generic path keeps the plain divisions: no per-face pass has vetted its faces)
    } else {
        const bool rev = fn >= p.F0;
        const float4* rv = reinterpret_cast<const float4*>(rv_b + (rev ? fn - p.F0 : fn));
        const float4 v0 = rv[0], v1 = rv[1], v2 = rv[2];
        v[0] = rev ? v1.z : v0.x; v[1] = rev ? v1.w : v0.y; v[2] = rev ? v2.x : v0.z;
        v[3] = v0.w; v[4] = v1.x; v[5] = v1.y;
        v[6] = rev ? v0.x : v1.z; v[7] = rev ? v0.y : v1.w; v[8] = rev ? v0.z : v2.x;
        return v2.y != 0.0f && !(p.dbg & 4096);  // division-safe (dbg 4096: plain divisions everywhere, for the A/B test)
    }
}

// the 9 floats of (virtual) face fn of image b
template <bool VC>
__device__ __forceinline__ void fetch_verts(const FwdParams& p, int b, int fn, float* v, int* vid) {
    if (!VC) {
        const float* g = p.faces + ((int64_t)b * p.F + fn) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = g[k];
    } else {
        const bool rev = fn >= p.F0;
        const int32_t* ix = p.fidx + ((int64_t)b * p.F0 + (rev ? fn - p.F0 : fn)) * 3;
        const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
        vid[0] = rev ? i2 : i0; vid[1] = i1; vid[2] = rev ? i0 : i2;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float* g = p.verts + ((int64_t)b * p.V + vid[k]) * 3;
            v[3 * k] = g[0]; v[3 * k + 1] = g[1]; v[3 * k + 2] = g[2];
        }
    }
}

__device__ __forceinline__ void zbuf_min(unsigned long long* zb, int idx, float zp, int fn) {
    const unsigned long long key = ((unsigned long long)f2ord(zp) << 32) | (unsigned)fn;
    atomicMin(&zb[idx], key);
}

// FUSED = true : write every pixel of every requested plane (fused epilogue).
// FUSED = false: upstream-compatible forward_face_index_map: touch hit pixels only.
#ifdef MR_WG_TIMELINE
// profiling builds (scripts/wg_timeline.py): per tile with geometry, start / end of its S1-S3 phase on the 100 MHz
// wall clock, candidate count, the compute unit it ran on and its workgroup
__device__ unsigned long long mr_dbg_times[65536 * 4];
#endif

// A tile no face can touch is pure background: every requested plane streamed with 16-byte stores (full tiles of rasters
// whose side is a multiple of 4; one 128-B row segment = 8 float4 of a scalar plane / 24 float4 of weight_map).
__device__ __forceinline__ void stream_background_tile(const FwdParams& p, int b, int t, int tx0, int ty0) {
    const int tid = threadIdx.x, is = p.is;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    // one 128-B row segment = 8 float4 (scalar planes) / 24 float4 (weight_map, 3 floats per pixel)
    const int64_t plane = (int64_t)is * is;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tid < TILE_H * 8) {
        const int row = tid >> 3, seg = tid & 7;
        const int64_t ro = ((int64_t)b * is + ty0 + row) * is + tx0;            // raster orientation
        const int64_t io = ((int64_t)b * is + (is - 1 - ty0 - row)) * is + tx0;  // image orientation
        const int m1 = -1;
        const float fm1 = __int_as_float(m1);
        reinterpret_cast<float4*>(p.fim + ro)[seg] = make_float4(fm1, fm1, fm1, fm1);
        if (p.depth && !p.sparse_wd) reinterpret_cast<float4*>(p.depth + io)[seg] = make_float4(p.far_, p.far_, p.far_, p.far_);
        if (p.alpha) reinterpret_cast<float4*>(p.alpha + io)[seg] = zero4;
        if (p.mask) reinterpret_cast<float4*>(p.mask + io)[seg] = zero4;
        if (p.rgb) {
            const float* bg = p.background + (int64_t)b * p.bg_stride;
            const int64_t o = ((int64_t)b * 3 * is + (is - 1 - ty0 - row)) * is + tx0;
#pragma unroll
            for (int c = 0; c < 3; c++)
                if (c < p.rgb_channels)
                    reinterpret_cast<float4*>(p.rgb + o + c * plane)[seg] = make_float4(bg[c], bg[c], bg[c], bg[c]);
        }
    }
    if (p.tile_hit && tid == 0) reinterpret_cast<uint32_t*>(p.tile_hit)[(int64_t)b * tiles_per_img + t] = 0u;
    if (p.weight && !p.sparse_wd && tid < TILE_H * 24) {
        const int row = tid / 24, seg = tid % 24;
        reinterpret_cast<float4*>(p.weight + (((int64_t)b * is + ty0 + row) * is + tx0) * 3)[seg] = zero4;
    }
}

// One screen tile `lid` (= image * tiles per image + tile): candidates -> z-buffer -> resolve (design at the top).
// `ent` (listed launches): the tile's list entry {lid, record offset, record count, large-list length}; else nullptr
template <bool FUSED, bool VC>
__device__ __forceinline__ void raster_one_tile(const FwdParams& p, const unsigned lid, const uint4* ent = nullptr) {
#ifdef MR_WG_TIMELINE
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    __shared__ unsigned long long zbuf[TILE_W * TILE_H];
    __shared__ float fcache[TPB / MR_WAVE][NB * FC_STRIDE];
    __shared__ unsigned short fragq[TPB / MR_WAVE][FQCAP];  // slot << 8 | row << 5 | x
    __shared__ float xp_tab[TILE_W], yp_tab[TILE_H];
    // the (face, row) items of a wave's batch, in face order: slot | tile row << 5 (round 6; until then a table of the faces' row
    // offsets that every item searched with five dependent LDS reads: -30 vector instructions per pass of 64 items)
    __shared__ unsigned char itemtab[TPB / MR_WAVE][NB * TILE_H];
    __shared__ int hitcnt[TPB / MR_WAVE];

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int b = lid / tiles_per_img;
    const int t = lid % tiles_per_img;
    const int tx0 = (t % p.tiles_x) * TILE_W, ty0 = (t / p.tiles_x) * TILE_H;
    const int tx1 = min(tx0 + TILE_W, p.is) - 1, ty1 = min(ty0 + TILE_H, p.is) - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int is = p.is;
    const float fis = (float)is;

    // Tiles the image's faces cannot touch (outside the union bbox of the live faces) are pure
    // background: stream it with 16-byte stores and leave before any LDS set-up.
    int n_rec = 0, n_bin = 0;
    int64_t off_bin = 0, off_large = 0;  // record indices relative to the image's record base
    if (!p.keys) {
        if (ent) {
            n_bin = (int)ent->z;
            off_bin = (int64_t)ent->y;
            n_rec = n_bin + (int)ent->w;
        } else {
            const BinHdr bh = p.bins[(int64_t)b * (p.nbx * p.nby) + ((t / p.tiles_x) >> p.ysh) * p.nbx + (t % p.tiles_x)];
            n_bin = (int)bh.cnt;
            off_bin = (int64_t)bh.off;
            n_rec = n_bin + p.hdrs[b].n_large;
        }
        off_large = (int64_t)SMALL_MAX_BINS * p.F - n_bin;  // so that record i >= n_bin sits at off_large + i
        if (p.dbg & 1) n_rec = 0;
        if (n_rec == 0 && (p.dbg & 256)) return;
        if (FUSED && n_rec == 0 && p.sparse_tiles) {
            if (tid == 0) reinterpret_cast<uint32_t*>(p.tile_hit)[(int64_t)b * tiles_per_img + t] = 0u;
            return;
        }
        if (FUSED && n_rec == 0 && (is & 3) == 0 && tx0 + TILE_W <= is && ty0 + TILE_H <= is && !p.face_inv_map) {
            stream_background_tile(p, b, t, tx0, ty0);
            return;
        }
    }

#ifdef MR_WG_TIMELINE
    if (threadIdx.x == 0 && lid < 65536u) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        mr_dbg_times[lid * 4 + 0] = dbg_t0;
        mr_dbg_times[lid * 4 + 2] = ((unsigned long long)n_rec << 32) | (xcc << 24) | (hwid & 0xffffffu);
        mr_dbg_times[lid * 4 + 3] = blockIdx.x;
    }
#endif
    zbuf[tid] = ~0ull;
    // NDC coordinates of the tile's pixel centres (upstream: (2 * i + 1 - is) / is)
    if (tid < TILE_W) xp_tab[tid] = (float)(2 * (tx0 + tid) + 1 - is) / fis;
    else if (tid < TILE_W + TILE_H) yp_tab[tid - TILE_W] = (float)(2 * (ty0 + tid - TILE_W) + 1 - is) / fis;
    __syncthreads();

    const FaceRec* recs_b = p.recs + (int64_t)b * REC_CAP * p.F;
    const RecVerts* rv_b = p.rverts + (int64_t)b * p.F0;
    // record i of this tile: the bin's list first, then the image's large list
    auto rec_at = [&](int i) -> const FaceRec* { return recs_b + (i < n_bin ? off_bin : off_large) + i; };
    float* fc = fcache[wave];
    unsigned short* fq = fragq[wave];
    unsigned char* itab = itemtab[wave];

    // S3: one lane per fragment -- barycentrics, near/far, depth test
    int fqh = 0, fqn = 0;  // fragment ring (wave-uniform)
    auto shade = [&](int n) {
        if (lane < n && !(p.dbg & 16)) {
            const unsigned fr = fq[(fqh + lane) & (FQCAP - 1)];
            const float* c = fc + (fr >> 8) * FC_STRIDE;
            const int lx = (int)(fr & 31u), ly = (int)((fr >> 5) & 7u);
            Face f;
#pragma unroll
            for (int k = 0; k < 9; k++) f.inv[k] = c[9 + k];
            f.v[2] = c[2]; f.v[5] = c[5]; f.v[8] = c[8];
            const int fslot = __float_as_int(c[18]);  // the face index
            float zp, w[3];
            const float yz[3] = {c[20], c[21], c[22]};
            if (!(c[23] != 0.0f && bary_shared(f, yz, tx0 + lx, ty0 + ly, zp, w))) bary(f, tx0 + lx, ty0 + ly, zp, w);
            // upstream: `if (zp <= near || far <= zp) continue; if (zp < depth) win`.  A NaN depth (faces whose
            // vertices coincide in x, y pass every edge test and have no inverse) survives the first test and
            // loses the second, so it must not reach the z-buffer -- both comparisons below are false for NaN
            if (zp > p.near_ && zp < p.far_) zbuf_min(zbuf, ly * TILE_W + lx, zp, fslot);
        }
        __builtin_amdgcn_wave_barrier();
    };

    auto process_batch = [&](int first, int count) {
        // S1: one lane per record
        int nrows = 0, row0 = 0;  // rows of the face's bbox inside this tile, the first of them
        if (lane < count && !(p.dbg & 4)) {
            const FaceRec r = *rec_at(first + lane);
            const int fn = (int)r.z;
            Face f;
            const bool safe = load_face_coords<VC>(p, rv_b, b, fn, f.v);
            float* c = fc + lane * FC_STRIDE;
            if (safe) {
                face_inverse_shared(f.v, f.inv, is);
                c[20] = rcp_refined(f.v[2]); c[21] = rcp_refined(f.v[5]); c[22] = rcp_refined(f.v[8]);
            } else {
                face_inverse(f.v, f.inv, is);
            }
            c[23] = safe ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < 9; k++) { c[k] = f.v[k]; c[9 + k] = f.inv[k]; }
            const int x0 = max((int)(r.x & 0xffffu), tx0) - tx0, x1 = min((int)(r.x >> 16), tx1) - tx0;
            const int y0 = max((int)(r.y & 0xffffu), ty0) - ty0, y1 = min((int)(r.y >> 16), ty1) - ty0;
            c[18] = __int_as_float(fn);
            c[19] = __int_as_float(x0 | (x1 << 8) | (y0 << 16) | (y1 << 24));
            // (a record of the image's large list, or of a bin taller than a tile, may miss this tile)
            nrows = (x0 <= x1 && y0 <= y1) ? y1 - y0 + 1 : 0;
            row0 = y0;
        }
        // (face, bbox row) items of the batch, compacted: exclusive prefix sum of the row counts -> the
        // S2 lanes take consecutive items, so a pass works on 64 real rows whatever the face sizes
        // (8 lanes per face would leave most of them idle for the 3 - 4-row faces of these meshes)
        const int incl = wave_incl_sum(nrows);  // (six DPP adds; the lanes beyond the batch hold 0)
        for (int r_ = 0; r_ < nrows; r_++) itab[incl - nrows + r_] = (unsigned char)(lane | ((row0 + r_) << 5));
        const int n_items = __builtin_amdgcn_readlane(incl, NB - 1);
        __builtin_amdgcn_wave_barrier();
        if (p.dbg & (4 | 8)) return;
        // S2: one lane per (face, bbox row) item, 64 items per pass.  Along a row each edge test
        //   reject_k(x) = ey_k < (xp[x] - a_k) * dy_k
        // is monotone in x even in floating point (xp[x] increases with x; IEEE subtraction and
        // multiplication by a constant are monotone), so the pixels a face covers on a row form ONE
        // span: three 6-step binary searches with the exact predicate find it -- the same pixels
        // the per-pixel loop would accept, without visiting the others.
        for (int i0 = 0; i0 < n_items; i0 += MR_WAVE) {
            const int item = i0 + lane;
            bool act = item < n_items;
            const unsigned it_ = act ? (unsigned)itab[item] : 0u;
            const int slot = (int)(it_ & 31u);
            const int row = (int)(it_ >> 5);
            float ea[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, ey[3] = {0, 0, 0};
            int lx0 = 0, lx1 = -1;
            if (act) {
                const float* c = fc + slot * FC_STRIDE;
                const int bb = __float_as_int(c[19]);
                const float ax = c[0], ay = c[1], bx = c[3], by = c[4], cx_ = c[6], cy_ = c[7];
                lx0 = bb & 0xff; lx1 = (bb >> 8) & 0xff;
                const float yp = yp_tab[row];
                ea[0] = ax; ea[1] = bx; ea[2] = cx_;
                dy[0] = by - ay; dy[1] = cy_ - by; dy[2] = ay - cy_;
                ey[0] = (yp - ay) * (bx - ax); ey[1] = (yp - by) * (cx_ - bx); ey[2] = (yp - cy_) * (ax - cx_);
            }
            int lo = lx0, hi = lx1;
            // T_k(x) = accepted_k(x) XOR (dy_k < 0) is prefix-true on [lx0, lx1] (accepted set of edge k:
            // a prefix if dy_k > 0 -- or dy_k == 0: all or nothing --, a suffix if dy_k < 0).  l_k = last x with
            // T_k true (lx0 - 1 if none) is bracketed by l < l_k + 1 <= h with the EXACT predicate only; where to
            // probe is free.  The first probe is the closed-form crossing of the edge with the row (approximate
            // reciprocal: it only has to land within a pixel), the next ones step to the neighbour of the bound
            // that just moved -- two or three probes settle almost every item -- and whatever is still open after
            // three probes is bisected.  The three edges advance together.
            int l[3], h[3], cc[3], tt[3];
            const bool dec[3] = {dy[0] < 0.0f, dy[1] < 0.0f, dy[2] < 0.0f};
            int probe[3];
            bool open_any = false;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                // (xp - a) dy = ey  ->  xp = a + ey / dy;  pixel index = (xp is + is - 1) / 2, tile-local
                const float xs = ((ea[k] + ey[k] * __builtin_amdgcn_rcpf(dy[k])) * fis + fis - 1.0f) * 0.5f - (float)tx0;
                // NaN (dy == 0) -> lx1: T is constant along the row then, the probe at the far end settles it
                int c = (int)fminf(fmaxf(floorf(xs), (float)lx0), (float)lx1);
                if (!(xs == xs)) c = lx1;
                // straight-line first round: the exact predicate at c and c + 1 (the crossing is almost always
                // between them); whatever this leaves open goes to the general bracketing loop below
                // (measured and dropped in round 6: for power-of-two rasters, the pixel centres by arithmetic -- (2 x + 1 - is) times the
                // exact reciprocal, the same bits -- instead of the two LDS reads that wait for c: 62.2 -> 63.1 us)
                const float xp0 = xp_tab[c & (TILE_W - 1)], xp1 = xp_tab[(c + 1) & (TILE_W - 1)];
                const bool t0 = (!(ey[k] < (xp0 - ea[k]) * dy[k])) != dec[k];
                const bool t1 = c < lx1 && ((!(ey[k] < (xp1 - ea[k]) * dy[k])) != dec[k]);
                l[k] = t0 ? c + (t1 ? 1 : 0) : lx0 - 1;
                // still open: T true at c + 1 < lx1 (the bound is further right) or false at c > lx0 (further left)
                open_any = open_any || (t0 ? (t1 && c + 1 < lx1) : c > lx0);
                cc[k] = c; tt[k] = (t0 ? 1 : 0) | (t1 ? 2 : 0);
            }
            if (!(p.dbg & 32) && __ballot(act && open_any) != 0ull) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const bool t0 = tt[k] & 1, t1 = tt[k] & 2;
                    h[k] = t0 ? (t1 ? lx1 + 1 : cc[k] + 1) : cc[k];
                    probe[k] = t0 ? cc[k] + 2 : cc[k] - 1;
                }
                for (int it = 1;; it++) {
                    open_any = false;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const bool go = h[k] - l[k] > 1;
                        const int mid = it < 3 ? min(max(probe[k], l[k] + 1), h[k] - 1) : (l[k] + h[k]) >> 1;
                        const float xp = xp_tab[max(mid, 0) & (TILE_W - 1)];
                        const bool t = (!(ey[k] < (xp - ea[k]) * dy[k])) != dec[k];
                        l[k] = (go && t) ? mid : l[k];
                        h[k] = (go && !t) ? mid : h[k];
                        probe[k] = t ? mid + 1 : mid - 1;
                        open_any = open_any || (h[k] - l[k] > 1);
                    }
                    if (__ballot(act && open_any) == 0ull) break;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (dec[k]) lo = max(lo, l[k] + 1); else hi = min(hi, l[k]);
            }
            int len = (act && !(p.dbg & 64)) ? max(hi - lo + 1, 0) : 0;
            int x = lo;
            // emit the spans as fragments, at most EMIT pixels per lane per round: a wave prefix sum gives every
            // lane its place in the ring
            constexpr int EMIT = 4;
            while (__ballot(len > 0) != 0ull) {
                const int c = min(len, EMIT);
                const int incl = wave_incl_sum(c);
                const int total = __builtin_amdgcn_readlane(incl, MR_WAVE - 1);
                const int at = fqh + fqn + incl - c;
                const unsigned tag = ((unsigned)slot << 8) | ((unsigned)row << 5);
#pragma unroll
                for (int i = 0; i < EMIT; i++)
                    if (i < c) fq[(at + i) & (FQCAP - 1)] = (unsigned short)(tag | (unsigned)(x + i));
                x += c; len -= c; fqn += total;
                __builtin_amdgcn_wave_barrier();
                while (fqn >= MR_WAVE) {
                    shade(MR_WAVE);
                    fqh = (fqh + MR_WAVE) & (FQCAP - 1);
                    fqn -= MR_WAVE;
                }
            }
        }
        if (fqn > 0) {
            shade(fqn);
            fqh = (fqh + fqn) & (FQCAP - 1);
            fqn = 0;
        }
    };

    // The tile's records go to the waves NB at a time, round robin (full S1 passes; a contiguous quarter per wave was
    // 3 us slower).  There is no candidate scan: the records of a bin overlap the bin by construction (a bin IS a tile
    // unless the image has more than MAX_BINS tiles), and the few that do not -- the image's large list, taller
    // bins -- get an empty row range in S1.
    for (int base = wave * NB; base < n_rec; base += (TPB / MR_WAVE) * NB) process_batch(base, min(NB, n_rec - base));
    const int lxr = tid & (TILE_W - 1), ly = tid >> 5;
    if (p.keys && tx0 + lxr < is && ty0 + ly < is) zbuf[tid] = p.keys[((int64_t)b * is + ty0 + ly) * is + tx0 + lxr];
    __syncthreads();

#ifdef MR_WG_TIMELINE
    if (threadIdx.x == 0 && lid < 65536u) mr_dbg_times[lid * 4 + 1] = wall_clock64();
#endif
    if (p.dbg & 2) return;
    // resolve: one pixel per thread (x = tid % 32, y = tid / 32)
    {
        const unsigned long long key = zbuf[tid];
        const int opx = tx0 + lxr, opy = ty0 + ly;  // this thread's own pixel: coverage bytes, background
        const bool hitpx = key != ~0ull && opx < is && opy < is;
        if (p.tile_hit) {
            const unsigned long long any = __ballot(hitpx);
            if (lane == 0) p.tile_hit[((int64_t)b * tiles_per_img + t) * 4 + wave] = any != 0ull ? 1 : 0;
        }
        const bool inside = opx < is && opy < is;
        if (inside && !hitpx) {
            const int64_t ri = ((int64_t)b * is + opy) * is + opx;            // raster orientation
            const int64_t ii = ((int64_t)b * is + (is - 1 - opy)) * is + opx;  // image orientation
            if (FUSED) {
                p.fim[ri] = -1;
                if (p.weight && !p.sparse_wd) { p.weight[ri * 3 + 0] = 0.0f; p.weight[ri * 3 + 1] = 0.0f; p.weight[ri * 3 + 2] = 0.0f; }
                if (p.depth && !p.sparse_wd) p.depth[ii] = p.far_;
                if (p.rec4) {
                    const float* bg = p.background + (int64_t)b * p.bg_stride;
                    p.rec4[ii] = make_float4(bg[0], bg[1], 0.0f, 0.0f);
                }
                if (p.alpha) p.alpha[ii] = 0.0f;
                if (p.mask) p.mask[ii] = 0.0f;
                if (p.rgb) {
                    const float* bg = p.background + (int64_t)b * p.bg_stride;
                    const int64_t plane = (int64_t)is * is;
                    const int64_t o = ((int64_t)b * 3 * is + (is - 1 - opy)) * is + opx;
                    p.rgb[o] = bg[0]; p.rgb[o + plane] = bg[1];
                    if (p.rgb_channels > 2) p.rgb[o + 2 * plane] = bg[2];
                }
                if (p.face_inv_map)
#pragma unroll
                    for (int k = 0; k < 9; k++) p.face_inv_map[ri * 9 + k] = 0.0f;
            }
        }
        // The covered pixels of the tile (two fifths of a tile with geometry, in runs of a few pixels per row) are
        // compacted before the expensive part -- the winner's set-up, barycentrics and sampling, ~180 instructions --
        // so that it runs in ceil(covered / 64) waves instead of in every wave that owns a covered pixel.  The list
        // lives in the (now idle) fragment ring of wave 0, the per-wave counts in its row-offset table.
        int q = tid;
        bool active = hitpx;
        if (!(p.dbg & 8192)) {
            unsigned short* hlist = fragq[0];
            int* hcnt = hitcnt;
            const unsigned long long m = __ballot(hitpx);
            if (lane == 0) hcnt[wave] = __popcll(m);
            __syncthreads();
            int hbase = 0, htotal = 0;
#pragma unroll
            for (int w = 0; w < TPB / MR_WAVE; w++) {
                const int c = hcnt[w];
                hbase += w < wave ? c : 0;
                htotal += c;
            }
            if (hitpx) hlist[hbase + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tid;
            __syncthreads();
            active = tid < htotal;
            q = active ? (int)hlist[tid] : tid;
        }
        if (!active) return;
        const unsigned long long wkey = zbuf[q];
        const int px = tx0 + (q & (TILE_W - 1)), py = ty0 + (q >> 5);
        const int64_t ri = ((int64_t)b * is + py) * is + px;            // raster orientation
        const int64_t ii = ((int64_t)b * is + (is - 1 - py)) * is + px;  // image orientation
        int fn = (int)(unsigned)(wkey & 0xffffffffull);
        const float zp = ord2f((uint32_t)(wkey >> 32));
        Face f;
        int vid[3] = {0, 0, 0};
        bool safe = false;
        float yz[3] = {0.0f, 0.0f, 0.0f};
        if (p.keys) {  // validation path: keys carry face indices, no binning pass ran
            fetch_verts<VC>(p, b, fn, f.v, vid);
            face_inverse(f.v, f.inv, is);
        } else {
            safe = load_face_coords<VC>(p, rv_b, b, fn, f.v);
            if (VC) {
                const bool rev = fn >= p.F0;
                const int32_t* ix = p.fidx + ((int64_t)b * p.F0 + (rev ? fn - p.F0 : fn)) * 3;
                const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
                vid[0] = rev ? i2 : i0; vid[1] = i1; vid[2] = rev ? i0 : i2;
            }
            if (safe) {
                face_inverse_shared(f.v, f.inv, is);
                yz[0] = rcp_refined(f.v[2]); yz[1] = rcp_refined(f.v[5]); yz[2] = rcp_refined(f.v[8]);
            } else {
                face_inverse(f.v, f.inv, is);
            }
        }
        // barycentrics of the winner, recomputed with the arithmetic of cover()
        float w[3], zp2;
        const bool shared_w = safe && bary_shared(f, yz, px, py, zp2, w);
        if (!shared_w) bary(f, px, py, zp2, w);
        (void)zp2;
        p.fim[ri] = fn;
        if (p.weight && !(VC && p.vid_map)) { p.weight[ri * 3 + 0] = w[0]; p.weight[ri * 3 + 1] = w[1]; p.weight[ri * 3 + 2] = w[2]; }
        if (p.face_inv_map)
#pragma unroll
            for (int k = 0; k < 9; k++) p.face_inv_map[ri * 9 + k] = f.inv[k];
        if (!FUSED) {
            p.depth[ri] = zp;
            return;
        }
        if (p.depth) p.depth[ii] = zp;
        if (p.alpha) p.alpha[ii] = 1.0f;
        float mask_v = 0.0f;
        if (p.mask || p.rec4) {  // flow_mask_kernel's arithmetic: (alpha > thresh) * (face + 1 inside the table ? lut : 1)
            float m = (1.0f > p.alpha_thresh) ? 1.0f : 0.0f;
            if (p.keep_lut) m = m * ((fn + 1 >= 0 && fn + 1 < p.n_lut) ? p.keep_lut[fn + 1] : 1.0f);
            if (p.mask) p.mask[ii] = m;
            mask_v = m;
        }
        if (p.rgb || p.rec4) {
            const int ts = p.ts;
            float tif[3];
            if (shared_w && mag_within(zp, -30, 30)) {  // tex_coords with depth / z_k through the depths' reciprocals
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float t = w[k] * (float)(ts - 1) * div_refined(zp, f.v[3 * k + 2], yz[k]);
                    t = fmaxf(t, 0.0f);
                    tif[k] = fminf(t, (float)(ts - 1) - p.eps);
                }
            } else {
                tex_coords(w, zp, f.v, ts, p.eps, tif);
            }
            float c[3] = {0.0f, 0.0f, 0.0f};
            if (!VC) {
                const float* tex = p.textures + ((int64_t)b * p.F + fn) * ts * ts * ts * 3;
#pragma unroll
                for (int pn = 0; pn < 8; pn++) {
                    float wg; int isc;
                    tex_tap(tif, pn, ts, wg, isc);
#pragma unroll
                    for (int k = 0; k < 3; k++) c[k] += wg * tex[isc * 3 + k];
                }
            } else {
                // the 2x2x2 vertex-colour texture of batch_vertex_textures, never materialised:
                // texel (1,0,0) = colour of vertex 0, (0,1,0) = vertex 1, (0,0,1) = vertex 2, else 0;
                // same 8-tap accumulation order as the texture path (bit-identical)
                // (tap axis k carries the colour of the face's vertex texel_vertex(k): the identity in the assumed layout)
                float vc[3][3];
                int cid[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    cid[k] = sel3(vid[0], vid[1], vid[2], texel_vertex(p.texel, k, fn >= p.F0));
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) vc[k][ch] = p.vcolors[((int64_t)b * p.V + cid[k]) * 3 + ch];
                }
                // (the five taps on zero texels are skipped: their weights are finite unless a sampling coordinate is
                // NaN, in which case the three taps below are NaN as well, and c + (+-0) == c for the sums at hand,
                // which start at +0 and therefore are never -0)
                float wgs[3];
#pragma unroll
                for (int pn = 1; pn <= 4; pn <<= 1) {
                    float wg = 1.0f;
#pragma unroll
                    for (int k = 0; k < 3; k++) wg *= ((pn >> k) & 1) ? (tif[k] - 0.0f) : (1.0f - (tif[k] - 0.0f));
                    wgs[pn >> 1] = wg;  // pn = 1, 2, 4 -> vertex 0, 1, 2
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float tv = (pn == 1) ? vc[0][ch] : (pn == 2) ? vc[1][ch] : vc[2][ch];
                        c[ch] += wg * tv;
                    }
                }
                if (p.vid_map) {
                    // (the three ids, then the three weights: two 12-byte stores -- interleaved, the two arrays' possible overlap
                    // kept the compiler at six 4-byte ones)
                    int32_t* vm = p.vid_map + ri * 3;
                    vm[0] = cid[0]; vm[1] = cid[1]; vm[2] = cid[2];
                    if (p.weight) {
                        float* wm = p.weight + ri * 3;
                        wm[0] = wgs[0]; wm[1] = wgs[1]; wm[2] = wgs[2];
                    }
                }
            }
            // rgb * mask + (1 - mask) * background with mask == 1 (rasterize.py:251-260)
            const float* bg = p.background + (int64_t)b * p.bg_stride;
            const int64_t plane = (int64_t)is * is;
            const int64_t o = ((int64_t)b * 3 * is + (is - 1 - py)) * is + px;
            if (p.rec4) p.rec4[ii] = make_float4(c[0] * 1.0f + 0.0f * bg[0], c[1] * 1.0f + 0.0f * bg[1], 1.0f, mask_v);
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (p.rgb && k < p.rgb_channels) p.rgb[o + k * plane] = c[k] * 1.0f + 0.0f * bg[k];
        }
    }
}

// (ListSlice / list_slice: mr_common.hpp)

// Listed launches whose tile list turned out LONGER than the grid (the caller's guess was low): the slice's entries
// beyond the first round, with the slice's stride.  Deliberately NOT inlined, and handed the kernel's argument block
// as a pointer into the kernarg segment: a loop around the tile body keeps every kernel argument live in scalar
// registers across its trips and spills 36 vector registers (175 instead of 142 us when the loop is the kernel, +2 us
// even as a never-entered second inlined copy of the body); as a called function its registers and spills are its own
// business, and nothing of it is executed when the guess held.  (A separate small looping launch for the overflow
// costs ~5 us on the timeline.)
template <bool FUSED, bool VC>
__device__ __attribute__((noinline)) void raster_overflow_tiles(const FwdParams* kernargs, const ListSlice sl, unsigned j) {
    const FwdParams& p = *kernargs;
    for (; j < sl.n_local; j += sl.stride) {
        __syncthreads();
        const uint4 ent = p.tile_ids[sl.slot(j, p.tile_cap)];
        raster_one_tile<FUSED, VC>(p, ent.x, &ent);
    }
}

// One tile per workgroup: workgroup i takes tile i (XCD-aware) or -- listed launches, p.tlist -- one entry of the list
// of tiles that hold candidates: the grid is sized by the caller's guess of the list length, no workgroup is dispatched
// for the empty 80 % of the screen and none pulls work through an atomic (a queue cursor serialised the launch,
// profiles/r02_persistent_tile_kernel_experiment.patch).  A list longer than the grid: raster_overflow_tiles.
template <bool FUSED, bool VC>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(7, 8))) raster_tile_kernel(FwdParams p) {
    // (ONE call site of the tile body: a second inlined copy costs the listed path 60 more scalar spills)
    const bool listed = p.tlist != nullptr;
    unsigned j = 0;
    ListSlice sl{};
    uint4 ent = make_uint4(xcd_remap(blockIdx.x, gridDim.x), 0u, 0u, 0u);
    bool have = true;
    if (listed) {
        const unsigned n_heavy = p.tlist->n_heavy, n_light = p.tlist->n_light;
        if (p.tile_count_out && blockIdx.x == 0 && threadIdx.x == 0) *p.tile_count_out = n_heavy + n_light;
        sl = list_slice(n_heavy, n_light, j);
        have = j < sl.n_local;
        if (have) ent = p.tile_ids[sl.slot(j, p.tile_cap)];
        if (FUSED && p.bg_ids) {
            // dense launch: this workgroup's share of the tiles no face touches -- stores only, in flight while the
            // tile's own candidates are worked on (the write stream of the empty four fifths of the screen rides on
            // the workgroups that have instructions to issue, instead of 13 000 workgroups of its own)
            // ... except those with a HEAVY tile: they are the launch's critical path (dispatched first for that reason)
            const unsigned n_bg = p.tlist->n_bg;
            const int tpi = p.tiles_x * p.tiles_y;
            const unsigned nx = (gridDim.x & 7u) ? 1u : 8u, nh_max = (n_heavy + nx - 1u) / nx;
            const bool few_heavy = nh_max * 2u <= sl.stride;  // (else: everybody helps)
            const unsigned k0 = few_heavy ? (j >= nh_max ? (j - nh_max) * nx + blockIdx.x % nx : n_bg) : blockIdx.x;
            const unsigned kstep = few_heavy ? (sl.stride - nh_max) * nx : gridDim.x;
            for (unsigned k = k0; k < n_bg; k += kstep) {
                const unsigned g = p.bg_ids[k];
                const int bt = (int)(g % (unsigned)tpi);
                stream_background_tile(p, (int)(g / (unsigned)tpi), bt, (bt % p.tiles_x) * TILE_W, (bt / p.tiles_x) * TILE_H);
            }
        }
    }
    if (have) raster_one_tile<FUSED, VC>(p, ent.x, listed ? &ent : nullptr);
    if (FUSED && listed && sl.n_local > sl.stride)  // (uniform per slice; false whenever the guess held)
        // (the kernel's own argument block, in the kernarg segment: C cast out of address space 4; the intrinsic is
        // only valid in the kernel itself -- inside the callee it read as a null pointer)
        raster_overflow_tiles<FUSED, VC>((const FwdParams*)__builtin_amdgcn_kernarg_segment_ptr(), sl, j + sl.stride);
}

// Validation-only variant (flags & MR_FLAG_REFERENCE_ALGO): upstream's structure, every pixel
// tests every face in ascending order with a strict '<'.  Writes the z-buffer key per pixel
// into `keys` (B*is*is u64), then the tile kernel's resolve is reused through a second launch.
__global__ void __launch_bounds__(256) raster_brute_kernel(const float* __restrict__ faces, int B, int F,
                                                           int is, float near_, float far_,
                                                           unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * is * is) return;
    const int b = (int)(i / ((int64_t)is * is));
    const int pn = (int)(i % ((int64_t)is * is));
    const int yi = pn / is, xi = pn % is;
    float depth_min = far_;
    int fmin_ = -1;
    for (int fn = 0; fn < F; fn++) {
        const float* g = faces + ((int64_t)b * F + fn) * 9;
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = g[k];
        if (backfacing(v)) continue;
        Face f;
#pragma unroll
        for (int k = 0; k < 9; k++) f.v[k] = v[k];
        face_inverse(f.v, f.inv, is);
        float zp, w[3];
        if (!cover(f, xi, yi, is, near_, far_, zp, w)) continue;
        if (zp < depth_min) { depth_min = zp; fmin_ = fn; }
    }
    keys[i] = (fmin_ >= 0) ? (((unsigned long long)f2ord(depth_min) << 32) | (unsigned)fmin_) : ~0ull;
}

// Upstream forward_texture_sampling on caller-provided maps (raster orientation, NHWC).
__global__ void __launch_bounds__(256) texture_sampling_kernel(
    const float* __restrict__ faces, const float* __restrict__ textures,
    const int32_t* __restrict__ fim, const float* __restrict__ weight_map,
    const float* __restrict__ depth_map, float* __restrict__ rgb_map,
    int32_t* __restrict__ sidx_map, float* __restrict__ swgt_map, int64_t npx, int F, int is, int ts,
    float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const int fn = fim[i];
    if (fn < 0) return;
    const int64_t bn = i / ((int64_t)is * is);
    const float* face = faces + (bn * F + fn) * 9;
    const float* tex = textures + (bn * F + fn) * ts * ts * ts * 3;
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = face[k];
    float w[3] = {weight_map[i * 3], weight_map[i * 3 + 1], weight_map[i * 3 + 2]};
    float tif[3];
    tex_coords(w, depth_map[i], v, ts, eps, tif);
    float c[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float wg; int isc;
        tex_tap(tif, pn, ts, wg, isc);
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] += wg * tex[isc * 3 + k];
        sidx_map[i * 8 + pn] = isc;
        swgt_map[i * 8 + pn] = wg;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) rgb_map[i * 3 + k] = c[k];
}

__global__ void __launch_bounds__(256) face_inv_map_kernel(const float* __restrict__ faces,
                                                           const int32_t* __restrict__ fim,
                                                           float* __restrict__ out, int64_t npx, int F, int is) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const int fn = fim[i];
    float inv[9];
    if (fn >= 0) {
        const int64_t b = i / ((int64_t)is * is);
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = faces[(b * F + fn) * 9 + k];
        face_inverse(v, inv, is);
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) inv[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) out[i * 9 + k] = inv[k];
}

// workspace layout: [B] ImageHdr | [B * nbins] BinHdr | [B * F] FaceBox | [B * REC_CAP * F] FaceRec | [B * F] RecVerts
// | TileList | [B * tiles] tile ids    (every byte the tile kernel reads is written by the two setup kernels: no memset)
struct WorkLayout {
    int nbx, nby, ysh;
    int parts;  // workgroups per image of the binning pass (bin_boxes_kernel, PARTS)
    size_t off_bins, off_boxes, off_recs, off_rverts, off_tlist, off_arrive, off_tile_ids, off_bg, off_part_cnt, total;
};

// Workgroups per image of the binning pass: the largest power of two (<= 16) that keeps B x parts within the device's
// compute units -- every workgroup of the launch resident at once, which the parts' barrier relies on -- and leaves a part
// at least 256 faces.  Phase stamps (scripts/wg_timeline.py, profiles/r05_binning_parts.txt): the exchange costs a part
// ~2.5 us (publish 0.4, barrier 0.8, the other parts' counters 1.4) once it stays inside the XCD's L2; 16 renders of 480 x 480
// (config 3, the reference's default batch size) 19.5 -> 13.9 us per workgroup, 64 renders 19.6 -> 16.6, 128 renders 21.7 -> ~19.
static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}
constexpr int MAX_PARTS = MAX_PARTS_DEV, MAX_PART_CUS = 256;
static int bin_parts(int B, int F, int cus, int per_cu = 2) {
    int k = 1;
    while (k < MAX_PARTS && (int64_t)B * k * 2 <= (int64_t)per_cu * cus && F / (k * 2) >= 256) k *= 2;
    return k;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static WorkLayout work_layout(int B, int F, int is) {
    WorkLayout w;
    const int tiles_x = (is + TILE_W - 1) / TILE_W, tiles_y = (is + TILE_H - 1) / TILE_H;
    w.nbx = tiles_x;
    w.ysh = 0;
    while ((int64_t)tiles_x * ((tiles_y + (1 << w.ysh) - 1) >> w.ysh) > MAX_BINS) w.ysh++;
    w.nby = (tiles_y + (1 << w.ysh) - 1) >> w.ysh;
    w.off_bins = align256((size_t)B * sizeof(ImageHdr));
    w.off_boxes = w.off_bins + align256((size_t)B * w.nbx * w.nby * sizeof(BinHdr));
    w.off_recs = w.off_boxes + align256((size_t)B * F * sizeof(FaceBox));
    w.off_rverts = w.off_recs + align256((size_t)B * REC_CAP * F * sizeof(FaceRec));
    w.off_tlist = w.off_rverts + align256((size_t)B * F * sizeof(RecVerts));  // (VC needs F / 2 of it with fill-back)
    // (the images' arrival counters sit right behind the list header: ONE region for the caller of MR_FLAG_TILE_LIST_CLEARED to clear)
    w.off_arrive = w.off_tlist + align256(sizeof(TileList));
    w.off_tile_ids = w.off_arrive + align256((size_t)B * ARRIVE_STRIDE * sizeof(unsigned));
    w.off_bg = w.off_tile_ids + align256((size_t)2 * B * tiles_x * tiles_y * sizeof(uint4));  // heavy part | light part
    w.off_part_cnt = w.off_bg + align256((size_t)B * tiles_x * tiles_y * sizeof(uint32_t));  // dense listed launches: ids of the tiles off the list
    // (sized for a 256-CU device, the most launch_bins assumes: the layout must not depend on the device at hand)
    w.parts = bin_parts(B, F, MAX_PART_CUS);
    w.total = w.off_part_cnt + (w.parts > 1 ? align256((size_t)B * w.parts * ((w.nbx * w.nby + 4 + 31) & ~31) * sizeof(int)) : 0);
    return w;
}

// per-face records + boxes (pass A), bin lists (pass B); fills the record-source fields of `fp`
// `tile_hit` != nullptr asks for the tile list too (sparse-tile launches of rasters whose bins are tiles): fp.tlist is
// set when one is built
// `dense_list`: build the list for a DENSE launch (every pixel written): the tiles without candidates are listed too, in
// an id array of their own, instead of getting zero coverage bytes
// `pro` (pair steps): the frame pair's vertex stage, to run inside the binning pass where that pass takes its fused-records
// form and the vertices fit its LDS (bin_boxes_kernel PROLOGUE) -- else as the launch of its own, in front, from here
template <bool VC>
static int launch_bins(BinParams bp, FwdParams& fp, void* workspace, int B, int F, int is, hipStream_t s,
                       uint8_t* tile_hit = nullptr, bool dense_list = false, bool list_cleared = false,
                       const PairPrologue* pro = nullptr) {
    const WorkLayout w = work_layout(B, F, is);
    if (w.nbx > MAX_BINS) return MR_ERR_BADARG;  // (image_size <= 16384 keeps a row of bins within the counters)
    char* base = (char*)workspace;
    if (dense_list) {
        bp.bg_ids = (uint32_t*)(base + w.off_bg);
        fp.bg_ids = bp.bg_ids;
        tile_hit = (uint8_t*)bp.bg_ids;  // (not written in this mode: any non-null value asks for the list)
    }
    if (tile_hit && w.ysh == 0 && (int64_t)B * w.nbx * w.nby <= 0x7fffffffLL) {
        bp.tlist = (TileList*)(base + w.off_tlist);
        bp.tile_ids = (uint4*)(base + w.off_tile_ids);
        bp.tile_hit = tile_hit;
        bp.tile_cap = (int64_t)B * w.nbx * w.nby;
        fp.tlist = bp.tlist; fp.tile_ids = bp.tile_ids; fp.tile_cap = (unsigned)bp.tile_cap;
    }
    bp.hdrs = (ImageHdr*)base;
    bp.bins = (BinHdr*)(base + w.off_bins);
    bp.boxes = (FaceBox*)(base + w.off_boxes);
    bp.recs = (FaceRec*)(base + w.off_recs);
    bp.rverts = (RecVerts*)(base + w.off_rverts);
    bp.F = F; bp.is = is; bp.nbx = w.nbx; bp.nby = w.nby; bp.ysh = w.ysh;
    const int nbins = w.nbx * w.nby;
    // parts per image (bin_boxes_kernel, PARTS): B x parts workgroups, all resident at once; dbg 32: one workgroup per image
    // (at most one workgroup per compute unit: two per compute unit -- dbg 64 -- were measured at 2B = 128: 37.7 us against
    // 28.9 with two parts and 28.5 with one; config 3's 16 renders: 16 parts 23.1 us, one part 26.0)
    int parts = (bp.dbg & 32) ? 1 : bin_parts(B, F, std::min(device_cus(), MAX_PART_CUS), 1);
    if (bp.dbg & 4) parts = std::max(1, parts / 2);  // (profiling: half / a quarter / an eighth of the parts)
    if (bp.dbg & 8) parts = std::max(1, parts / 4);
    if (parts > w.parts) parts = w.parts;
    // (list_cleared: the caller cleared the tile list's header on this stream -- what the per-face pass's first thread does)
    const bool fused_records = VC && list_cleared && bp.tlist && bp.F0 > 0 && !(bp.dbg & 16);
    // The parts pay where they split the PER-FACE pass (fused_records: 64 renders of 256 x 256 22.1 us with 4 parts against 24.3
    // with one, 16 renders of 480 x 480 21.9 / 24.5, 64 of 640 x 640 31.1 / 35.5; 128 renders: 28.3 / 28.5).  With the boxes
    // already in memory a part only saves counting, and the exchange costs more than that unless the bins are many: 128 renders
    // of 256 x 256 19.6 us with two parts, 17.8 with one; 64 renders 16.9 / 15.6; 16 of 480 x 480 18.3 / 17.5; 64 of 640 x 640
    // (1600 bins) 26.2 / 27.9 (round 6, rocprofv3 over `bench.py --kernels-only` / scripts/hot_only.py, HOC_FWD_DBG sweeps).
    if (!fused_records && nbins <= 1024 && !(bp.dbg & 1)) parts = 1;
    bp.B = B; bp.parts = parts;
    bp.part_cnt = (int*)(base + w.off_part_cnt);
    bp.arrive = (unsigned*)(base + w.off_arrive);
    size_t lds = (size_t)((nbins + 3) & ~3) * sizeof(int);
    // (boxes of the largest part: its share of the real faces in both orientations, or of the virtual faces)
    int box_cap = (F + parts - 1) / parts + 2;
    if (pro) {  // (the parts of a pair step split at the hand / object boundary: pair_part_range)
        int most = 0;
        for (int k = 0; k < parts; k++) {
            int r0, nr;
            pair_part_range(k, parts, pro->f.Fh, pro->f.Fo, r0, nr);
            most = std::max(most, nr);
        }
        box_cap = std::max(box_cap, (bp.fill_back ? 2 * most : most) + 2);
    }
    bp.box_cap = box_cap;
    bp.split_fh = pro ? pro->f.Fh : 0; bp.split_fo = pro ? pro->f.Fo : 0;
    const size_t box_lds = (size_t)box_cap * sizeof(FaceBox);
    bp.lds_boxes = (lds + box_lds <= 152 * 1024) ? 1 : 0;
    if (bp.lds_boxes) lds += box_lds;
    fp.hdrs = bp.hdrs; fp.bins = bp.bins; fp.recs = bp.recs; fp.rverts = bp.rverts;
    fp.nbx = w.nbx; fp.nby = w.nby; fp.ysh = w.ysh;
    if (B == 0) return MR_OK;
    if (B > 65535) return MR_ERR_BADARG;
    const bool fused_records_ok = fused_records && bp.lds_boxes;
    bool pro_fused = false;
    if (pro) {
        const size_t v_lds = (size_t)bp.V * 3 * sizeof(float);
        if (VC && fused_records_ok && (B & 1) == 0 && lds + v_lds <= 152 * 1024 && !(bp.dbg & 128)) {
            pro_fused = true;
            lds += v_lds;
        } else {
            const int rc = mr_launch_pair_prologue(*pro, s);
            if (rc != MR_OK) return rc;
        }
    }
    if (fused_records_ok) {
        // nothing: the per-face pass runs inside the binning kernel
    } else if (bp.F0 > 0) {
        hipLaunchKernelGGL((face_records_kernel<VC>), dim3((unsigned)((bp.F0 + 255) / 256), (unsigned)B), dim3(256), 0, s,
                           bp);
        MR_CHECK_LAUNCH();
    } else if (bp.tlist) {
        // no face, no per-face pass: the list counters that pass clears (thread 0 of image 0) are cleared here, before
        // the binning pass adds to them -- an empty mesh gives an empty list, not whatever the workspace held
        hipError_t e = hipMemsetAsync(bp.tlist, 0, sizeof(TileList), s);
        if (e != hipSuccess) return (int)e;
    }
    static size_t allowed = 48 * 1024;  // dynamic LDS beyond the default limit is an opt-in, raised on demand
    if (lds > allowed) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bin_boxes_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bin_boxes_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        const void* more[] = {reinterpret_cast<const void*>(&bin_boxes_prologue_kernel), reinterpret_cast<const void*>(&bin_count_prologue_kernel),
                              reinterpret_cast<const void*>(&bin_count_kernel<false>), reinterpret_cast<const void*>(&bin_count_kernel<true>),
                              reinterpret_cast<const void*>(&bin_fill_kernel<false>), reinterpret_cast<const void*>(&bin_fill_kernel<true>)};
        for (const void* f : more)
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return (int)e;
        allowed = 152 * 1024;
    }
    // (parts > 1: workgroup i = part (i / 8) % parts of image (i / 8 / parts) * 8 + i % 8 -- an image's parts on one XCD)
    const unsigned grid = parts > 1 ? (unsigned)((B + 7) / 8) * 8u * (unsigned)parts : (unsigned)B;
    // (eight parts or more: the parts' exchange across a kernel boundary -- count launch, fill launch.  Measured on one box, pair
    // steps, device time per hot-path pass: 16 renders of 480 x 480 in 16 parts 27.7 us as one launch, 13.0 + 8.7 as two (0.097 ->
    // 0.092 ms); 32 renders in 8 parts and 64 in 4: no difference (0.0811 / 0.0818, 0.1109 / 0.1112); 128 renders of 256 x 256 in
    // 2 parts 33.4 against 22.4 + 13.1 -- with few parts the last arriver's serial share is small and the second launch's floor
    // is not.  dbg 64: the last-arriver form whatever the parts)
    const bool two_launches = parts >= 8 && !(bp.dbg & 64);
    if (two_launches) {
        if (pro_fused) hipLaunchKernelGGL(bin_count_prologue_kernel, dim3(grid), dim3(BIN_TPB), lds, s, bp, *pro);
        else if (fused_records_ok) hipLaunchKernelGGL(bin_count_kernel<true>, dim3(grid), dim3(BIN_TPB), lds, s, bp);
        else hipLaunchKernelGGL(bin_count_kernel<false>, dim3(grid), dim3(BIN_TPB), lds, s, bp);
        MR_CHECK_LAUNCH();
        const size_t lds2 = (size_t)((nbins + 3) & ~3) * sizeof(int);  // (counters only: the fill launch keeps no boxes in LDS)
        if (fused_records_ok) hipLaunchKernelGGL(bin_fill_kernel<true>, dim3(grid), dim3(BIN_TPB), lds2, s, bp);
        else hipLaunchKernelGGL(bin_fill_kernel<false>, dim3(grid), dim3(BIN_TPB), lds2, s, bp);
    } else if (pro_fused) hipLaunchKernelGGL(bin_boxes_prologue_kernel, dim3(grid), dim3(BIN_TPB), lds, s, bp, *pro);
    else if (fused_records_ok) hipLaunchKernelGGL(bin_boxes_kernel<true>, dim3(grid), dim3(BIN_TPB), lds, s, bp);
    else hipLaunchKernelGGL(bin_boxes_kernel<false>, dim3(grid), dim3(BIN_TPB), lds, s, bp);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

// `tile_bound` (listed launches): the caller's guess of the tile-list length = workgroups to dispatch; a longer list
// is walked with a grid stride, surplus workgroups leave after one scalar load
template <bool FUSED, bool VC>
static int launch_tiles(FwdParams& p, hipStream_t s, int64_t tile_bound = 0) {
    if (p.rgb_channels == 0) p.rgb_channels = 3;
    p.tiles_x = (p.is + TILE_W - 1) / TILE_W;
    p.tiles_y = (p.is + TILE_H - 1) / TILE_H;
    int64_t nblocks = (int64_t)p.B * p.tiles_x * p.tiles_y;
    if (nblocks == 0) return MR_OK;
    if (nblocks > 0x7fffffffLL) return MR_ERR_BADARG;
    if (p.tlist) {
        // Listed launch: `bound` workgroups take one list entry each; whatever the list holds beyond the caller's guess
        // is walked afterwards by the same workgroups (raster_overflow_tiles).
        if (tile_bound <= 0) tile_bound = (nblocks + 3) / 4;
        const int64_t bound = std::min<int64_t>(nblocks, std::max<int64_t>((tile_bound + 7) & ~(int64_t)7, 8));
        if constexpr (FUSED) {
            hipLaunchKernelGGL((raster_tile_kernel<FUSED, VC>), dim3((unsigned)bound), dim3(TPB), 0, s, p);
            MR_CHECK_LAUNCH();
            return MR_OK;
        }
        return MR_ERR_NOTIMPL;  // (no lists for the five-entry-point compatible path)
    }
    hipLaunchKernelGGL((raster_tile_kernel<FUSED, VC>), dim3((unsigned)nblocks), dim3(TPB), 0, s, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

// mr_selftest_division: the shared-reciprocal division primitives next to the compiler's `/`, element by element
__global__ void __launch_bounds__(256) selftest_division_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                float* __restrict__ refined, float* __restrict__ plain,
                                                                int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    refined[i] = div_refined(a[i], b[i], rcp_refined(b[i]));
    plain[i] = a[i] / b[i];
}

}  // namespace mr

using namespace mr;

// Can a dense launch (every requested plane written for every pixel) go over the tile list?  Full tiles with 16-byte rows,
// no per-pixel inverse map (the background stream does not write one), a raster whose bins are tiles.
static bool dense_list_ok(int B, int F, int is, const float* face_inv_map, int flags) {
    if ((flags & (MR_FLAG_REFERENCE_ALGO | MR_FLAG_TILE_PER_WORKGROUP)) || face_inv_map) return false;
    if ((is % TILE_W) != 0 || (is % TILE_H) != 0) return false;
    const WorkLayout w = work_layout(B, F, is);
    return w.ysh == 0 && (int64_t)B * w.nbx * w.nby <= 0x7fffffffLL;
}

// ... over the tile list: a quarter of the screen as the grid (a longer list is walked in rounds, surplus workgroups leave
// after their share of the background), every workgroup streams its share of the tiles off the list first
template <bool VC>
static int launch_dense_listed(FwdParams& p, hipStream_t s) {
    if (!p.bg_ids) return MR_ERR_BADARG;
    return launch_tiles<true, VC>(p, s, -1);
}

extern "C" int mr_selftest_division(const float* a, const float* b, float* refined, float* plain, int64_t n,
                                    mr_stream_t stream) {
    if (n < 0 || (n > 0 && (!a || !b || !refined || !plain))) return MR_ERR_BADARG;
    if (n == 0) return MR_OK;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return MR_ERR_BADARG;
    hipLaunchKernelGGL(selftest_division_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, refined, plain, n);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int64_t mr_render_workspace_bytes(int batch_size, int num_faces, int image_size) {
    if (batch_size < 0 || num_faces < 0 || image_size <= 0 || image_size > 16384) return MR_ERR_BADARG;
    return (int64_t)work_layout(batch_size, num_faces, image_size).total;
}

extern "C" int64_t mr_render_clear_bytes(int batch_size, int num_faces, int image_size) {
    if (batch_size < 0 || num_faces < 0 || image_size <= 0 || image_size > 16384) return MR_ERR_BADARG;
    const WorkLayout w = work_layout(batch_size, num_faces, image_size);
    return (int64_t)(w.off_tile_ids - w.off_tlist);  // list header + the images' arrival counters (a multiple of 256)
}

extern "C" int mr_render_tile_list(const void* workspace, int batch_size, int num_faces, int image_size,
                                   const void** list_header, const void** list_entries, int64_t* list_capacity) {
    if (!workspace || !list_header || !list_entries || !list_capacity) return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0 || image_size > 16384) return MR_ERR_BADARG;
    const WorkLayout w = work_layout(batch_size, num_faces, image_size);
    // (a list is built when a bin is a tile and the global tile ids fit 31 bits: launch_bins)
    if (w.ysh != 0 || (int64_t)batch_size * w.nbx * w.nby > 0x7fffffffLL) return MR_ERR_NOTIMPL;
    const char* base = (const char*)workspace;
    *list_header = base + w.off_tlist;
    *list_entries = base + w.off_tile_ids;
    *list_capacity = (int64_t)batch_size * w.nbx * w.nby;
    return MR_OK;
}

extern "C" int mr_forward_face_index_map(const float* faces, int32_t* face_index_map, float* weight_map,
                                         float* depth_map, float* face_inv_map, float* faces_inv,
                                         int batch_size, int num_faces, int image_size, float near_,
                                         float far_, int return_rgb, int return_alpha, int return_depth,
                                         mr_stream_t stream) {
    (void)return_rgb; (void)return_alpha;
    if (!faces || !face_index_map || !weight_map || !depth_map || !faces_inv) return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0 || image_size > 16384) return MR_ERR_BADARG;
    if (return_depth && !face_inv_map) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    hipStream_t s = (hipStream_t)stream;
    void* work = nullptr;
    const size_t bytes = (size_t)mr_render_workspace_bytes(batch_size, num_faces, image_size) + 256;
    hipError_t e = hipMallocAsync(&work, bytes, s);
    if (e != hipSuccess) return (int)e;
    FwdParams p{};
    BinParams bp{};
    bp.faces = faces; bp.faces_inv = faces_inv; bp.F0 = num_faces;
    int rc = launch_bins<false>(bp, p, work, batch_size, num_faces, image_size, s);
    if (rc == MR_OK) {
        p.faces = faces;
        p.depth = depth_map; p.fim = face_index_map;
        p.weight = weight_map; p.face_inv_map = return_depth ? face_inv_map : nullptr;
        p.B = batch_size; p.F = num_faces; p.is = image_size; p.ts = 1;
        p.near_ = near_; p.far_ = far_; p.eps = 0.0f;
        rc = launch_tiles<false, false>(p, s);
    }
    e = hipFreeAsync(work, s);
    if (rc == MR_OK && e != hipSuccess) rc = (int)e;
    return rc;
}

extern "C" int mr_forward_texture_sampling(const float* faces, const float* textures,
                                           const int32_t* face_index_map, const float* weight_map,
                                           const float* depth_map, float* rgb_map,
                                           int32_t* sampling_index_map, float* sampling_weight_map,
                                           int batch_size, int num_faces, int image_size,
                                           int texture_size, float eps, mr_stream_t stream) {
    if (!faces || !textures || !face_index_map || !weight_map || !depth_map || !rgb_map ||
        !sampling_index_map || !sampling_weight_map)
        return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0 || texture_size < 2) return MR_ERR_BADARG;
    const int64_t npx = (int64_t)batch_size * image_size * image_size;
    if (npx == 0) return MR_OK;
    hipLaunchKernelGGL(texture_sampling_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, faces, textures, face_index_map, weight_map, depth_map,
                       rgb_map, sampling_index_map, sampling_weight_map, npx, num_faces, image_size,
                       texture_size, eps);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_render_forward(const float* faces, const float* textures, const float* background,
                                 int bg_stride, float* rgb_img, float* alpha_img, float* depth_img,
                                 int32_t* face_index_map, float* weight_map, float* face_inv_map,
                                 void* workspace, int64_t workspace_bytes, int batch_size,
                                 int num_faces, int image_size, int texture_size, float near_,
                                 float far_, float eps, int return_rgb, int return_alpha,
                                 int return_depth, int flags, mr_stream_t stream) {
    if ((!faces && num_faces > 0) || !face_index_map || !weight_map || !workspace) return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0 || image_size > 16384) return MR_ERR_BADARG;
    if (return_rgb && (!rgb_img || (!textures && num_faces > 0) || !background || texture_size < 2))
        return MR_ERR_BADARG;
    if (return_rgb && bg_stride != 0 && bg_stride != 3) return MR_ERR_BADARG;
    if (return_alpha && !alpha_img) return MR_ERR_BADARG;
    if (return_depth && !depth_img) return MR_ERR_BADARG;
    if (workspace_bytes < mr_render_workspace_bytes(batch_size, num_faces, image_size)) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    hipStream_t s = (hipStream_t)stream;
    FwdParams p{};
    BinParams bp{};
    bp.faces = faces; bp.F0 = num_faces;
    bp.dbg = (flags >> 24) & 0xff;
    const bool dense_list = dense_list_ok(batch_size, num_faces, image_size, face_inv_map, flags);
    int rc = (flags & MR_FLAG_REFERENCE_ALGO) ? MR_OK
                                              : launch_bins<false>(bp, p, workspace, batch_size, num_faces, image_size, s, nullptr, dense_list);
    if (rc != MR_OK) return rc;
    p.faces = faces;
    p.textures = textures; p.background = background;
    p.bg_stride = bg_stride;
    p.rgb = return_rgb ? rgb_img : nullptr;
    p.alpha = return_alpha ? alpha_img : nullptr;
    p.depth = return_depth ? depth_img : nullptr;
    p.fim = face_index_map; p.weight = weight_map;
    p.face_inv_map = face_inv_map;
    p.B = batch_size; p.F = num_faces; p.is = image_size; p.ts = return_rgb ? texture_size : 1;
    p.near_ = near_; p.far_ = far_; p.eps = eps;
    p.dbg = flags >> 8;
    if (flags & MR_FLAG_REFERENCE_ALGO) {
        const int64_t npx = (int64_t)batch_size * image_size * image_size;
        unsigned long long* keys = nullptr;
        hipError_t e = hipMallocAsync((void**)&keys, (size_t)npx * 8 + 256, s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(raster_brute_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, faces,
                           batch_size, num_faces, image_size, near_, far_, keys);
        rc = (int)hipGetLastError();
        p.keys = keys;
        if (rc == MR_OK) rc = launch_tiles<true, false>(p, s);
        e = hipFreeAsync(keys, s);
        if (rc == MR_OK && e != hipSuccess) rc = (int)e;
        return rc;
    }
    if (p.tlist) return launch_dense_listed<false>(p, s);
    return launch_tiles<true, false>(p, s);
}

extern "C" int mr_face_inv_map(const float* faces, const int32_t* face_index_map, float* face_inv_map,
                               int batch_size, int num_faces, int image_size, mr_stream_t stream) {
    if (!faces || !face_index_map || !face_inv_map) return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0) return MR_ERR_BADARG;
    const int64_t npx = (int64_t)batch_size * image_size * image_size;
    if (npx == 0) return MR_OK;
    hipLaunchKernelGGL(face_inv_map_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, faces, face_index_map, face_inv_map, npx, num_faces, image_size);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_render_vc_forward(const float* verts, const int32_t* faces_idx, const float* vcolors,
                                    const float* background, int bg_stride, float* rgb_img, float* alpha_img,
                                    float* depth_img, int32_t* face_index_map, float* weight_map,
                                    void* workspace, int64_t workspace_bytes, int batch_size, int num_verts,
                                    int num_faces, int fill_back, int image_size, float near_, float far_,
                                    float eps, int return_rgb, int return_alpha, int return_depth, int flags,
                                    int texel_layout, mr_stream_t stream) {
    const int F = fill_back ? 2 * num_faces : num_faces;
    if (!texel_layout_ok(texel_layout)) return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || num_verts < 0 || image_size <= 0 || image_size > 16384) return MR_ERR_BADARG;
    if (((!verts || !faces_idx) && num_faces > 0) || !face_index_map || !weight_map || !workspace) return MR_ERR_BADARG;
    if (return_rgb && (!rgb_img || (!vcolors && num_faces > 0) || !background || !(eps >= 1e-6f))) return MR_ERR_BADARG;
    if (return_rgb && bg_stride != 0 && bg_stride != 3) return MR_ERR_BADARG;
    if ((return_alpha && !alpha_img) || (return_depth && !depth_img)) return MR_ERR_BADARG;
    if (workspace_bytes < mr_render_workspace_bytes(batch_size, F, image_size)) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    if (batch_size > 65535) return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    FwdParams p{};
    BinParams bp{};
    bp.verts = verts; bp.fidx = faces_idx; bp.V = num_verts; bp.F0 = num_faces; bp.fill_back = fill_back;
    bp.dbg = (flags >> 24) & 0xff;
    if (batch_size > 65535) return MR_ERR_BADARG;
    const bool dense_list = dense_list_ok(batch_size, F, image_size, nullptr, flags);
    const int rc = launch_bins<true>(bp, p, workspace, batch_size, F, image_size, s, nullptr, dense_list);
    if (rc != MR_OK) return rc;
    p.background = background; p.bg_stride = bg_stride;
    p.rgb = return_rgb ? rgb_img : nullptr;
    p.alpha = return_alpha ? alpha_img : nullptr;
    p.depth = return_depth ? depth_img : nullptr;
    p.fim = face_index_map; p.weight = weight_map;
    p.B = batch_size; p.F = F; p.is = image_size; p.ts = 2;
    p.near_ = near_; p.far_ = far_; p.eps = eps;
    p.verts = verts; p.fidx = faces_idx; p.vcolors = vcolors; p.V = num_verts; p.F0 = num_faces;
    p.texel = texel_layout;
    p.dbg = (flags >> 8) & 0xffff;  // profiling experiments (scripts/fwd_vc_variants.py)
    if (p.dbg & 128) return MR_OK;  // ... binning pass alone
    if (p.tlist) return launch_dense_listed<true>(p, s);
    return launch_tiles<true, true>(p, s);
}

#ifdef MR_WG_TIMELINE
extern "C" __attribute__((visibility("default"))) int mr_debug_times(void* dst, long n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mr::mr_dbg_times), n, 0, hipMemcpyDeviceToHost);
}
extern "C" __attribute__((visibility("default"))) int mr_debug_bin_times(void* dst, long n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mr::mr_dbg_bin), n, 0, hipMemcpyDeviceToHost);
}
#endif

// mr_render_flow_forward of the 2B stacked meshes of a frame pair whose vertex stage has NOT run yet (mr_pair_step_forward,
// pair_step.hip): `verts` (unused where the stage runs inside the binning pass), `faces_idx` and `vcolors` are the buffers that
// stage fills -- pro->v.ndc1 / cols12 and pro->f.out point into them
int mr_render_flow_forward_pair(const float* verts, const int32_t* faces_idx, const float* vcolors,
                                const float* background, int bg_stride, const float* keep_lut, int n_lut,
                                float alpha_thresh, float* rgb_img, float* alpha_img, float* mask_img,
                                float* depth_img, float* weight_map, int32_t* face_index_map, uint8_t* tile_hit,
                                void* workspace, int64_t workspace_bytes, int batch_size,
                                int num_verts, int num_faces, int fill_back, int image_size, float near_,
                                float far_, float eps, int flags, int32_t* vertex_id_map, int tile_bound,
                                uint32_t* tile_count_out, float* zero_fill, int64_t zero_fill_count,
                                int texel_layout, mr_stream_t stream, const mr::PairPrologue* pro, void* records);

extern "C" int mr_render_flow_forward(const float* verts, const int32_t* faces_idx, const float* vcolors,
                                      const float* background, int bg_stride, const float* keep_lut, int n_lut,
                                      float alpha_thresh, float* rgb_img, float* alpha_img, float* mask_img,
                                      float* depth_img, float* weight_map, int32_t* face_index_map, uint8_t* tile_hit,
                                      void* workspace, int64_t workspace_bytes, int batch_size,
                                      int num_verts, int num_faces, int fill_back, int image_size, float near_,
                                      float far_, float eps, int flags, int32_t* vertex_id_map, int tile_bound,
                                      uint32_t* tile_count_out, float* zero_fill, int64_t zero_fill_count,
                                      int texel_layout, mr_stream_t stream) {
    return mr_render_flow_forward_pair(verts, faces_idx, vcolors, background, bg_stride, keep_lut, n_lut, alpha_thresh, rgb_img,
                                       alpha_img, mask_img, depth_img, weight_map, face_index_map, tile_hit, workspace,
                                       workspace_bytes, batch_size, num_verts, num_faces, fill_back, image_size, near_, far_, eps,
                                       flags, vertex_id_map, tile_bound, tile_count_out, zero_fill, zero_fill_count, texel_layout,
                                       stream, nullptr, nullptr);
}

int mr_render_flow_forward_pair(const float* verts, const int32_t* faces_idx, const float* vcolors,
                                const float* background, int bg_stride, const float* keep_lut, int n_lut,
                                float alpha_thresh, float* rgb_img, float* alpha_img, float* mask_img,
                                float* depth_img, float* weight_map, int32_t* face_index_map, uint8_t* tile_hit,
                                void* workspace, int64_t workspace_bytes, int batch_size,
                                int num_verts, int num_faces, int fill_back, int image_size, float near_,
                                float far_, float eps, int flags, int32_t* vertex_id_map, int tile_bound,
                                uint32_t* tile_count_out, float* zero_fill, int64_t zero_fill_count,
                                int texel_layout, mr_stream_t stream, const mr::PairPrologue* pro, void* records) {
    const int F = fill_back ? 2 * num_faces : num_faces;
    if (!texel_layout_ok(texel_layout)) return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || num_verts < 0 || image_size <= 0 || image_size > 16384) return MR_ERR_BADARG;
    if (((!verts || !faces_idx || !vcolors) && num_faces > 0) || !face_index_map || !workspace) return MR_ERR_BADARG;
    if (vertex_id_map && !weight_map) return MR_ERR_BADARG;
    // (`records`: [B,is,is] 16-byte records {colour 0, colour 1, alpha, mask} in place of the three planes -- listed sparse
    // launches only: the dense background stream writes planes)
    if (records ? (rgb_img || alpha_img || mask_img || ((uintptr_t)records & 15) || !(flags & MR_FLAG_SPARSE_TILES) || tile_bound == 0)
                : (!rgb_img || !alpha_img || !mask_img))
        return MR_ERR_BADARG;
    if (!background || !(eps >= 1e-6f)) return MR_ERR_BADARG;
    if ((bg_stride != 0 && bg_stride != 3) || (keep_lut && n_lut <= 0)) return MR_ERR_BADARG;
    if (workspace_bytes < mr_render_workspace_bytes(batch_size, F, image_size)) return MR_ERR_BADARG;
    if (zero_fill_count < 0 || (zero_fill_count > 0 && !zero_fill)) return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (batch_size == 0) {
        if (zero_fill && zero_fill_count > 0) return (int)hipMemsetAsync(zero_fill, 0, (size_t)zero_fill_count * sizeof(float), s);
        return MR_OK;
    }
    if (batch_size > 65535) return MR_ERR_BADARG;
    if (pro && ((batch_size & 1) || pro->v.B * 2 != batch_size || pro->v.V != num_verts || pro->f.Fh + pro->f.Fo != num_faces))
        return MR_ERR_BADARG;
    FwdParams p{};
    BinParams bp{};
    bp.verts = verts; bp.fidx = faces_idx; bp.V = num_verts; bp.F0 = num_faces; bp.fill_back = fill_back;
    bp.dbg = (flags >> 24) & 0xff;
    bp.zero_fill = zero_fill_count > 0 ? zero_fill : nullptr; bp.zero_count = zero_fill_count;
    if ((flags & MR_FLAG_SPARSE_TILES) && !tile_hit) return MR_ERR_BADARG;
    const bool listed = (flags & MR_FLAG_SPARSE_TILES) && tile_bound != 0;
    const int rc = launch_bins<true>(bp, p, workspace, batch_size, F, image_size, s, listed ? tile_hit : nullptr, false,
                                     (flags & MR_FLAG_TILE_LIST_CLEARED) != 0, pro);
    if (rc != MR_OK) return rc;
    p.tile_count_out = p.tlist ? tile_count_out : nullptr;
    p.background = background; p.bg_stride = bg_stride;
    p.rgb = rgb_img; p.rgb_channels = 2;
    p.alpha = alpha_img; p.mask = mask_img; p.rec4 = (float4*)records;
    if (records && !p.tlist) return MR_ERR_NOTIMPL;  // (no tile list for this raster: pair_step_layout has asked before)
    p.depth = depth_img; p.weight = weight_map; p.sparse_wd = 1; p.tile_hit = tile_hit; p.vid_map = vertex_id_map;
    p.sparse_tiles = (flags & MR_FLAG_SPARSE_TILES) ? 1 : 0;
    if (p.sparse_tiles && !tile_hit) return MR_ERR_BADARG;
    p.keep_lut = keep_lut; p.n_lut = n_lut; p.alpha_thresh = alpha_thresh;
    p.fim = face_index_map;
    p.B = batch_size; p.F = F; p.is = image_size; p.ts = 2;
    p.near_ = near_; p.far_ = far_; p.eps = eps;
    p.verts = verts; p.fidx = faces_idx; p.vcolors = vcolors; p.V = num_verts; p.F0 = num_faces;
    p.texel = texel_layout;
    p.dbg = (flags >> 8) & 0xffff;
    if (p.dbg & 128) return MR_OK;
    return launch_tiles<true, true>(p, s, tile_bound);
}
