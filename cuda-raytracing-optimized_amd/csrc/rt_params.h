// rt_params.h — kernel argument blocks and the launch entry points each kernel TU exports.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_types.h"

// Row partition of the image over devices / processes: the image is cut into
// stripes of `stripe_rows` rows; stripe k belongs to part (k % world).  A device renders its
// stripes into a COMPACT buffer of `local_rows` rows; local row lr maps to global row
//   j = ((lr / stripe_rows) * world + rank) * stripe_rows + lr % stripe_rows.
// Pixel seeds always use the GLOBAL pixel id, so the assembled image does not depend on the partition.
// Spheres per group of the sphere kernel (slots [g * kSphereGroup, (g + 1) * kSphereGroup) share one bounding box); 16 or 32.
// Measured on C2: a ray enters ~1.1 group boxes whatever their size (8: 1.34, 16: 1.32, 32: 1.19 for primary rays), so the
// sphere tests per ray grow with the group while the box tests shrink: 16 -> 5100 Msamples/s, 32 -> 4950.
constexpr int kSphereGroupShift = 4;
constexpr int kSphereGroup = 1 << kSphereGroupShift;
constexpr int kCellCount = 64;                          // cells per axis of the group boxes' cell tables (RtSphereParams::cell_on)
constexpr int kCellWordsMax = 8;                        // the tables hold one bit per group in up to 8 words per cell: scenes of up to 256 groups (4096 spheres)
// Words per cell for a scene of n_groups groups (0 = no tables: too many groups).  The tables - 6 x kCellCount x words words = rt_cell_f4 float4 - lie behind
// the 3 x n_groups box entries of RtSphereParams::groups and are staged into the LDS with them; host and kernels size them with this one function.
__host__ __device__ inline int rt_cell_words(int n_groups) { const int w = (n_groups + 31) / 32; return w <= kCellWordsMax ? w : 0; }
// float4 entries of the tables: 3 axes x 2 kinds (begins / ends) x kCellCount cells x words
__host__ __device__ inline int rt_cell_f4(int n_groups) { return (6 * kCellCount * rt_cell_words(n_groups)) / 4; }

struct RtPartition {
    int32_t stripe_rows;
    int32_t rank;
    int32_t world;
    int32_t local_rows;
};

struct RtCounters {             // device-side, optional
    unsigned long long rays;
    unsigned long long prim_tests;
    unsigned long long node_visits;
    unsigned long long shadow_rays;
    unsigned long long exec_tests;   // sphere tests actually executed per lane in the lane-parallel scan (after group culling)
    unsigned long long box_tests;    // sphere scenes: group / node box slab tests executed
    unsigned long long ref_stats[RT_STAT_COUNT];   // the reference's STATS counters, kernels.cu:47-67
};

struct RtSphereParams {
    rt_camera cam;
    int32_t nx, ny, ns, max_depth;
    int32_t n;                  // spheres
    int32_t n_padded;           // slots: multiple of 64; slots are grouped 16 by 16 (pad slots can never be hit)
    int32_t n_groups;           // n_padded / 16
    int32_t n_big_groups;       // groups [0, n_big_groups) hold the big spheres: always scanned
    int32_t n_big;              // real big spheres: slots [0, n_big)
    const float4* spheres;      // the kernel's sphere image: slot k at index k + k/16 (one pad entry per group of 16: LDS banks), (cx, cy, cz, radius*radius);
                                // n_padded + n_groups entries, spatially sorted (see rt_renderer.hip build_sphere_groups)
    const float*  rad;          // n_padded: radius of slot k (the hit normal divides by it, intersections.h:95)
    int32_t fb_global_rows;     // 1 = `fb` is the WHOLE image (the pinned host framebuffer, written over the bus as pixels finish): rows are global, not local
    const struct RtSphereParams* self;   // a device copy of this struct (launcher): the kernel re-reads what it needs once per sample / pixel from it
    int32_t basic_materials;    // 1 = every material is RT_DIFFUSE, RT_METAL or RT_GLASS (no look preset): the lean shading instantiation may run (material_scatter<BASIC>)
    int32_t global_scene;       // 1 = the scene does not fit the LDS: the kernels read these arrays from global memory (L2) instead of staging them
    const float4* groups;       // 2 x n_groups: inflated AABB (lo.xyz, hi.xyz) of each group
    // per-ray culling margin (exactness of the culling for ANY ray origin, see make_box_ray): centre and radius of the
    // small spheres' centres, k1 = K eps / (2 r_min), k2 = sqrt(K eps), k3 = slab-test rounding per unit of coordinate,
    // coord_max = largest |coordinate| of any group box
    float cull_cx, cull_cy, cull_cz, cull_radius, cull_k1, cull_k2, cull_k3, cull_coord_max;
    float box_shared_lo, box_shared_hi;   // that extent
    int32_t box_shared_axis;    // 1 + axis on which every (non-empty) group box has the same extent (spheres resting on a plane), 0 = none: see group_needs
    float pair_k0;              // 2 * kPairSlack * (largest radius of the grouped spheres)^2: see the pair rounds of scan_pairs
    // Cell tables of the group boxes (scenes of up to 32 x kCellWordsMax groups; see group_needs_cells): behind the 3 x n_groups box entries of `groups` lie, for
    // each of the three axes, kCellCount cells over the boxes' extent on that axis and per cell two sets of rt_cell_words(n_groups) words each (bit g of word
    // g / 32 = small group g): the boxes that begin at or below the cell, the boxes that end at or above it.  Index: (((2 axis + kind) kCellCount + cell) words
    // + word.  cell = x * cell_scale[axis] + cell_off[axis].  cell_axes: bit a = axis a's table can reject something (not every box has the same extent there:
    // spheres resting on a horizontal plane leave bit 1 clear).  ubox: the union of the group boxes (lo.xyz, hi.xyz) the ray is clipped to first.
    int32_t cell_on;
    int32_t cell_axes;
    float cell_scale[3], cell_off[3];
    float ubox[6];
    const float4* mat_color;    // n_padded x (r, g, b, param)
    const int32_t* mat_type;    // n_padded
    const int32_t* orig;        // n_padded: caller's sphere index of the slot, INT_MAX for pad slots
    const int32_t* slot_of;     // n: slot of the caller's sphere index
    rt_vec3* fb;                // compact framebuffer: local_rows x nx
    RtPartition part;
    int32_t sky;
    int32_t rr;
    int32_t rng_mode;
    float   t_min;
    RtCounters* counters;       // nullptr = off
    uint32_t* queue;            // one zero-initialised word: next unassigned pixel (persistent-wave kernels)
    unsigned long long* wave_dbg;   // nullptr, or 8 x u64 per wave: diagnostic time stamps (RT_WAVE_DEBUG)
    uint32_t* order;            // 3 * padded pixel count words: work-order lists built by the classify pre-pass
    int32_t spw;                // samples per work item (= ns in the reference-stream mode: one item per pixel)
    int32_t chunks;             // work items per pixel = ceil(ns / spw)
    rt_vec3* partial;           // chunks > 1: local_rows * nx * chunks un-normalised partial sums
    // two-phase rendering of the reference-stream mode (see rt_kernels_spheres.hip, "cost-ordered second phase"):
    int32_t phase;              // 0 = single launch; 1 = first samples [0, s_split) -> per-pixel state; 2 = resume [s_split, ns)
    int32_t s_split;            // samples rendered by phase 1
    int32_t chain_top_thr;      // 16 x rays per sample from which a pixel goes to chain list 0 (the longest chains; see kChainClasses)
    int32_t poison_fb;          // 1 = `fb` is the host framebuffer (fb_global_rows) and this frame must start from NaN in the rows of this partition member: the
                                // frame's first dispatch (which stores no pixel) fills them, or a small kernel in front of a single dispatch (poison_rows)
    float4* px_state;           // local_rows * nx: (col.xyz, rng bits) after phase 1
    uint32_t* px_rays;          // local_rows * nx: rays traced by phase 1
    float4* ord_state;          // the same two, copied by the ordering pass into QUEUE order (index = position in `order`): a lane that fetches a pixel in phase 2
    uint32_t* ord_rays;         // reads its list entry, state and ray count with three independent loads (by pixel they hang behind the list entry: one more round trip)
    // Traffic forms of the two-dispatch frame (round 4, NS-2; sphere launcher only - the mesh launcher leaves all three at 0):
    float4* ord_rec;            // != nullptr: the ordering pass writes ONE 32-byte record per queue position instead of the three arrays above - (col.xyz, rng bits |
                                // row << 16 | column, rays, 0, 0) at ord_rec[2 pos], [2 pos + 1]: one sector per pixel written and read, not three.  (Measured and dropped:
                                // 24-byte records as three 8-byte pieces - 8 MB less per frame, the frame 4 % slower.)
    int32_t xcd_queues;         // 8: one set of cost lists and queue counters per XCD (P.queue + 64 x, kXcdQueueWords); a pixel belongs to the XCD
                                // rt_xcd_of_pixel(local row, nx, column) - a hash of (local row * nx + column) >> 5 -, so the 32 pixels of three adjacent 128-byte framebuffer lines are finished by waves behind ONE
                                // L2, which merges their 12-byte stores into whole lines (`fb` in device memory).  0 / 1: one queue for the machine
    int32_t p1_tile_major;      // 1: the first dispatch hands out its two-sample items in tile-major order (a wave parks the pixels of two adjacent 8x8 tiles: its 16-byte
                                // states fill whole lines) instead of scattered
};
constexpr int kXcdQueueWords = 64;      // words per queue block: [0] general counter [1] chain counter [2] middle-tier counter [3] first position of this XCD's lists in
                                        // `order` [4 .. 4 + 18) list lengths [22 .. 22 + 18) fill cursors
constexpr int kXcdQueues = 8;
// (the unit index hashed, not its low bits: an image 256 k pixels wide - 3840 - would give every XCD the same columns of every row)
__host__ __device__ inline uint32_t rt_xcd_of_pixel(uint32_t local_row, uint32_t nx, uint32_t column) {
#ifdef RT_XCD_PLAIN
    return ((local_row * nx + column) >> 5) & 7u;
#else
    return ((((local_row * nx + column) >> 5) * 2654435761u) >> 29) & 7u;
#endif
}

struct RtMeshParams {
    rt_camera cam;
    int32_t nx, ny, ns, max_depth;
    const rt_triangle* tris;
    const float4* bvh4;         // heap-indexed nodes viewed as float4 texels: child pair of node i at texels 3i..3i+2
    const float* bvh_axis;      // the same child pairs, one 96-byte record per internal node i, grouped by AXIS: for a in x,y,z the eight floats
                                // {min_L, min_R, max_L, max_R | max_L, max_R, min_L, min_R} (L = node 2i, R = node 2i+1): a ray reads one float4 per
                                // axis, the first (1/dir >= 0) or the second (1/dir < 0), and gets (near_L, near_R, far_L, far_R) - the swap of
                                // intersections.h:30 done by the address (default kernel only)
    const float4* leaf_tri;     // compact leaf records for the pair rounds of the default kernel: per triangle SLOT (leaf x nppl + k) 3 float4 = (v0.xyz, e1.x | e1.yz, e2.xy |
                                // e2.z, -, -, -) with e1 = v1 - v0, e2 = v2 - v0 rounded as intersections.h:56-57 rounds them; sentinel slots are zero and never read;
                                // nullptr = not built (a leaf with a real triangle behind a sentinel)
    const uint32_t* leaf_ofs;   // number of real triangles of every leaf, one BYTE per leaf (packed four to a word; staged into the LDS by the kernel)
    uint32_t first_leaf;
    uint32_t nppl;
    int32_t leaf_sentinels_trailing;   // host-checked: no real triangle behind a sentinel in any leaf (pair rounds allowed)
    int32_t lean_ok;            // host-checked: every material is RT_DIFFUSE / RT_METAL / RT_GLASS and untextured (the lean instantiation may run)
    rt_bbox bounds;
    const rt_material* materials;
    const float* const* tex_data;
    const int32_t* tex_width;
    const int32_t* tex_height;
    rt_vec3* fb;
    RtPartition part;
    int32_t sky;
    int32_t nee;
    int32_t rr;
    int32_t rng_mode;
    float   t_min;
    rt_sphere light;
    rt_vec3 lightColor;
    int32_t floor_on;           // rt_render_options.floor: rays that miss the mesh are tested against `floor` (kernels.cu:341-345 re-enabled)
    rt_plane floor;             // kernel_scene.floor
    RtCounters* counters;
    uint32_t* queue;
    unsigned long long* dbg;    // diagnostics (RT_WAVE_DEBUG): 16 phase counters summed over all waves, or nullptr
    // cost-ordered frame in two dispatches (reference RNG stream; rt_launch_mesh_*): the first renders samples [0, s_split) of every pixel and parks (colour sum,
    // stream position) and the rays it took; rt_order_pixels_by_cost sorts the pixels by that cost into queue order; the second resumes them longest first
    int32_t s_split;
    float4* px_state;           // local_rows * nx
    uint32_t* px_rays;          // local_rows * nx
    uint32_t* order;            // queue position -> (local row << 16 | column)
    float4* ord_state;          // the parked state in queue order
    uint32_t* ord_rays;
    // the traffic forms of the two-dispatch frame, as in RtSphereParams (ord_rec, xcd_queues; p1_segments = its p1_tile_major == 2)
    float4* ord_rec;
    int32_t xcd_queues;
    int32_t p1_segments;
};

// LDS the sphere kernel needs for a scene of n spheres (n_padded slots).
size_t rt_sphere_kernel_lds_bytes(int n_padded, int n);
// The ordering pass of the two-dispatch frames (rt_kernels_spheres.hip, k_order_by_cost; also used by the mesh launcher): reads px_rays / px_state / s_split /
// chain_top_thr / nx / part.local_rows of `q`, writes order / ord_state / ord_rays (all valid pixels, descending cost class, scattered inside a class) and the
// list lengths into q.queue[4..]; q.queue must have been zeroed on `stream` before.
hipError_t rt_order_pixels_by_cost(const RtSphereParams& q, hipStream_t stream);

// Each returns the hipError_t of the launch.  `variant` selects a kernel variant (0 = default).
// Sphere launchers: p.self must point to a device copy of `p` that is complete on `stream` before the launch (the renderer owns it).
hipError_t rt_launch_spheres_parity(const RtSphereParams& p, int variant, hipStream_t stream);
hipError_t rt_launch_spheres_fast(const RtSphereParams& p, int variant, hipStream_t stream);
hipError_t rt_launch_mesh_parity(const RtMeshParams& p, int variant, hipStream_t stream);
hipError_t rt_launch_mesh_fast(const RtMeshParams& p, int variant, hipStream_t stream);
