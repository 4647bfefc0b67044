// rtmi.hip -- gfx950 (MI355X) kernels and the C ABI of include/rtmi.h.
//
// Replaces the reference's raytraceWorld / raytracePixels / calculateLighting call tree
// (RayTraceUI/RayTracingFunctions.swift:59-114, :145-271) with a staged GPU pipeline:
//
//   k_frame_tables, k_pack_tree   per camera: sample tables; cones of the triangles and of the
//                                 tree nodes above them (rtmi_tree.h)
//   k_gen               per sample: FP32 pre-test, ray generation, sphere/light tests (FP64)
//   k_search_tree       per wave of 256 rays: one walk of the cone tree, packed-FP32 cone and
//                       edge tests; emits (sample, list position) candidates
//   k_search_exact      per candidate: the reference's triangle test in FP64; lowest list
//                       position per ray
//   k_shade_candidates  per candidate: the ray's intersection is shaded (calculateLighting, FP64)
//   k_shade             per queued hit (spheres, lights, top-level triangles): the same shading
//   k_resolve           per pixel: sample average, x255 truncation, RGBA8 store
//
// RTMI_SEARCH_LIST_SCAN replaces the three search kernels by k_search_primary (k_pack_flags /
// _scan / _write): every front-facing triangle in list order until the ray's first hit, one lane
// per 4 rays, triangle operands wave-uniform in SGPRs, two triangles per v_pk_fma_f32.
// Scenes that list triangles at top level add k_search_finish / k_resolve_top (primary rays) and
// walk a bounding-sphere tree for their reflection and shadow rays (top_tree_trace).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rtmi.h"
#include "rtmi_exact.h"
#include "rtmi_shade32.h"

using rtx::d3;
using rtx::TriPlane;

typedef float f2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void rtmi_set_last_error(const std::string& msg) { g_last_error = msg; }  // used by rtmi_obj.cpp

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                     ":" + std::to_string(__LINE__) + ")";                                 \
      return e_ == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP;            \
    }                                                                                      \
  } while (0)

// No C++ exception leaves the C ABI: a failed host allocation becomes RTMI_ERR_OUT_OF_MEMORY.
template <class F>
static int32_t guarded(F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    g_last_error = "out of host memory";
    return RTMI_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    g_last_error = std::string("unexpected failure: ") + e.what();
    return RTMI_ERR_HIP;
  } catch (...) {
    g_last_error = "unexpected failure";
    return RTMI_ERR_HIP;
  }
}

// Inline-asm constraint for "keep this value in a scalar register"; a build of the kernels for another
// instruction set (the CPU build under tests/emu) names a register class of its own.
#ifndef RTMI_SCALAR_REG
#define RTMI_SCALAR_REG "+s"
#endif

#include "kernels_common.h"
#include "kernels_prepare.h"
#include "kernels_secondary.h"
#include "kernels_gen.h"
#include "kernels_scan.h"
#include "kernels_tree.h"
#include "kernels_shade.h"
#include "kernels_fast.h"

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// Device memory a frame needs beyond the scene itself.  Per sample of a batch: outcome, queues, ray
// direction, intersection parameter, colour, lowest list position; the candidate lists of the tree
// search; the per-column and per-row sample tables; the list of 8x8 pixel blocks.
// RTMI_NORMAL_CONES=0: secondary rays do not use the cones of normals of the bounding-sphere tree (for comparison).
static bool normal_cones_enabled() {
  static bool v = [] {
    const char* e = getenv("RTMI_NORMAL_CONES");
    return !(e && atoi(e) == 0);
  }();
  return v;
}
struct WorkBufs {
  size_t cap_samples = 0;
  int* hit_id = nullptr;
  int* tri_hit = nullptr;
  int* q_child = nullptr;
  int* q_shade = nullptr;
  int* q_top = nullptr;
  double* rgb = nullptr;
  double* dirs = nullptr;
  double* tval = nullptr;
  int* best_pos = nullptr;   // tree search: per sample, lowest list position hit
  int2* cand = nullptr;      // tree search: candidate pairs, cap_cand entries in NSLOT lists
  size_t cap_cand = 0;
  int2* multi = nullptr;     // tree search: the candidates of rays that have several, cap_multi entries in NSLOT lists
  size_t cap_multi = 0;
  double* tabs = nullptr;    // utab then vtab (FP64), then their FP32 copies
  size_t cap_tabs = 0;
  int2* blocks = nullptr;
  unsigned char* block_live = nullptr;  // per block of the list: can any of its samples hit anything (k_block_cull)
  size_t cap_blocks = 0;
  DevCounters* cnt = nullptr;  // work counters, queue and list fill levels
};

// Work buffers of the frames that run without a ray-generation launch (kernels_fast.h), grown on demand like the others.
// Per sample of a batch 24 bytes (outcome, lowest list position, colour with its bound) plus the candidate lists.
struct FastBufs {
  size_t cap_samples = 0;
  int* hit_id = nullptr;
  int* best_pos = nullptr;
  float4* rgbf = nullptr;
  int2* cand = nullptr;
  size_t cap_cand = 0;
  int2* multi = nullptr;
  size_t cap_multi = 0;
  int* fix_list = nullptr;   // pixels for k_fix_pixels
  size_t cap_fix = 0;
  int2* live_blocks = nullptr;  // the batch's live 8x8 blocks (k_block_cull_compact); cap_samples = cap_live x 64 x spp
  int* entries = nullptr;       // per live block: ENTRY_K start records of its waves' walks (k_block_entries)
  size_t cap_live = 0;          // blocks
  int* block_slot = nullptr;    // per block of the host's list: slot among the live ones, or -1
  size_t cap_slots = 0;
  size_t live_wanted = 0;       // blocks: what the arrays are grown to when a frame last overflowed them
  unsigned char* consts = nullptr;  // per frame: LightF per light, MatF per material, SphereF per object
  size_t cap_consts = 0;
  std::vector<unsigned char> consts_uploaded;  // what `consts` holds (the async copy reads it after the call returns)
};

// ---------------------------------------------------------------------------------------
// static work buffers of rtmi_render_device_noalloc
// ---------------------------------------------------------------------------------------
// Callers that cannot allocate device memory (code under graph capture, serving loops, harnesses
// that forbid it) render small frames with the work buffers below instead of buffers from
// hipMalloc.  Every device the library runs on holds its own copy of the array, at its own
// address, until the process ends; g_scratch_ptr[device] is that address (hipGetSymbolAddress with
// the device current, at the first rtmi_scene_create on it).  Host code never uses the array's
// name for anything else: the name is a host-side handle, not a device address.
constexpr size_t NOALLOC_MAX_SAMPLES = 65536;                                  // samples of one frame (whole 8x8 blocks x spp)
constexpr size_t NOALLOC_CAND = 2 * NOALLOC_MAX_SAMPLES + (size_t)NSLOT * 4096;  // candidate entries of the tree search
constexpr size_t NOALLOC_TAB_ENTRIES = 16384;                                  // (width + height) x sub-samples per axis
constexpr size_t NOALLOC_BLOCKS = NOALLOC_MAX_SAMPLES / BLK_PIX;
constexpr size_t SCRATCH_BYTES = 9u << 20;
constexpr int MAX_DEVICES = 64;

static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

static std::mutex g_scratch_mu;
static unsigned char* g_scratch_ptr[MAX_DEVICES];
// held for the whole of a rtmi_render_device_noalloc call: the calls of one device share its array, so they take turns
static std::mutex g_scratch_busy[MAX_DEVICES];

constexpr size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr size_t scratch_bytes_needed() {
  return 6 * align256(sizeof(int) * NOALLOC_MAX_SAMPLES) + 2 * align256(sizeof(double) * 3 * NOALLOC_MAX_SAMPLES) +
         align256(sizeof(double) * NOALLOC_MAX_SAMPLES) + align256(sizeof(int2) * NOALLOC_CAND) +
         align256((sizeof(double) + sizeof(float)) * 3 * NOALLOC_TAB_ENTRIES) + align256(sizeof(int2) * NOALLOC_BLOCKS) +
         align256(NOALLOC_BLOCKS) + align256(sizeof(DevCounters));
}
static_assert(scratch_bytes_needed() <= SCRATCH_BYTES, "the static array is too small for its slices");

// The slices of a device's static array, in the shape the render path expects.
static WorkBufs carve_scratch(unsigned char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* p = base + off;
    off += align256(bytes);
    return p;
  };
  WorkBufs w;
  w.cap_samples = NOALLOC_MAX_SAMPLES;
  w.hit_id = (int*)take(sizeof(int) * NOALLOC_MAX_SAMPLES);
  w.tri_hit = (int*)take(sizeof(int) * NOALLOC_MAX_SAMPLES);
  w.q_child = (int*)take(sizeof(int) * NOALLOC_MAX_SAMPLES);
  w.q_shade = (int*)take(sizeof(int) * NOALLOC_MAX_SAMPLES);
  w.q_top = (int*)take(sizeof(int) * NOALLOC_MAX_SAMPLES);
  w.best_pos = (int*)take(sizeof(int) * NOALLOC_MAX_SAMPLES);
  w.rgb = (double*)take(sizeof(double) * 3 * NOALLOC_MAX_SAMPLES);
  w.dirs = (double*)take(sizeof(double) * 3 * NOALLOC_MAX_SAMPLES);
  w.tval = (double*)take(sizeof(double) * NOALLOC_MAX_SAMPLES);
  w.cand = (int2*)take(sizeof(int2) * NOALLOC_CAND);
  w.cap_cand = NOALLOC_CAND;
  w.tabs = (double*)take((sizeof(double) + sizeof(float)) * 3 * NOALLOC_TAB_ENTRIES);
  w.cap_tabs = 3 * NOALLOC_TAB_ENTRIES;
  w.blocks = (int2*)take(sizeof(int2) * NOALLOC_BLOCKS);
  w.block_live = (unsigned char*)take(NOALLOC_BLOCKS);
  w.cap_blocks = NOALLOC_BLOCKS;
  w.cnt = (DevCounters*)take(sizeof(DevCounters));
  return w;
}

// Address of `scratch` on `device`, which must be the current device.
static int32_t scratch_init(int device) {
  if (device < 0 || device >= MAX_DEVICES) return RTMI_OK;  // rtmi_render_device_noalloc is refused there
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[device]) return RTMI_OK;
  void* p = nullptr;
  HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  g_scratch_ptr[device] = static_cast<unsigned char*>(p);
  return RTMI_OK;
}

struct TriGroup {
  int object;          // owning object index (bounding sphere), or -1 for the top-level runs
  int n = 0;           // triangles in the group
  int* d_group = nullptr;       // ordinal -> triangle index
  unsigned char* d_keep = nullptr;
  int* d_block_counts = nullptr;
  float* d_cone = nullptr;      // (n/8 + 2) batches of CONE_DWORDS
  float* d_edge = nullptr;      // n x EDGE_FLOATS
  int* d_packed_tri = nullptr;
  int* d_npacked = nullptr;     // triangles kept by the last pack
  size_t cone_bytes = 0;
  // cone tree (rtmi_tree.h): spatial slot order fixed at scene creation, cones rebuilt per frame
  TreeDev tree{};
  int* d_slot_tri = nullptr;
  int* d_slot_pos = nullptr;
  float* d_rec = nullptr;
  float* d_tedge = nullptr;
  NodeCone* d_nodes = nullptr;
  unsigned int* d_ticket = nullptr;
  size_t n_records = 0;  // records of the tree, all levels
  TreeHost tree_host;    // host copy of the slot order, kept until the scene is built
  // which operands the device holds: 0 none, 1 cone tree, 2 list scan -- and for which camera position
  int packed_for = 0;
  double packed_cam[3] = {0, 0, 0};
  double packed_err = 0.0;      // the direction error the tree's margins were widened by (k_pack_tree)
  float4* d_pos_plane = nullptr;  // per list position: the triangle's plane for the FP32 lighting (k_pack_tree)
  int2* d_pos_mt = nullptr;       // per list position: material index, triangle index
};

struct rtmi_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  std::vector<rtmi_object> objects;
  int n_lights = 0, n_materials = 0;
  long long n_tris = 0;
  rtmi_object* d_objects = nullptr;
  rtmi_light* d_lights = nullptr;
  rtmi_material* d_materials = nullptr;
  rtmi_triangle* d_tris = nullptr;
  TriPlane* d_planes = nullptr;
  int* d_tri_owner = nullptr;
  float* d_trec = nullptr;      // bounding-sphere records of the top-level runs (secondary rays)
  float* d_gst2 = nullptr;      // their per-triangle plane and edge operands
  TopTreeArgs top_args{};
  // RTMI_FLAG_SHADOWS_ENTER_BOUNDED: a second tree of the same kind over the children of all bounding spheres,
  // built by the first frame that asks for it (ensure_bounded_tree)
  bool bounded_tree_built = false;
  int* d_bmembers = nullptr;    // member ordinal -> triangle index
  int* d_bowner = nullptr;      // member ordinal -> object index of its bounding sphere
  int* d_bslot_pos = nullptr;
  float* d_brec = nullptr;
  float* d_bgst2 = nullptr;
  ExtArgs bounded_args{};       // tree, member count, coordinate centre (flags are set per frame)
  std::vector<TriGroup> groups;  // [0] = top-level runs (may be empty), then bounding objects
  SceneDev dev{};
  // per-batch work buffers: the scene's own (grown on demand), or for the duration of one
  // rtmi_render_device_noalloc call slices of the library's static device array
  WorkBufs wb;
  FastBufs fast;
  bool no_alloc = false;  // the frame being enqueued must fit the buffers as they are
  std::vector<SearchArgs> h_sargs;   // one argument block per triangle group
  SearchArgs* d_sargs = nullptr;
  std::array<double, 18> tabs_key{};  // what the tables were computed for
  bool tabs_valid = false;
  std::vector<SearchArgs> sargs_uploaded;  // what d_sargs holds
  float4* d_genf = nullptr;                // k_gen pre-test operands, one per sphere-like object
  std::vector<float4> genf_uploaded;
  std::vector<rtmi_light> h_lights;
  std::vector<rtmi_material> h_materials;
  std::vector<int2> h_blocks;          // host copy of the block list last uploaded
  std::array<int, 9> blocks_key{};     // frame geometry + tile selection it was built for
  bool blocks_valid = false;
  int n_plain_shadeable = 0;  // spheres and lights in the object list (hits on them go through the shade queue)
  // whether the lean shading of k_shade_candidates can hand a sample to the shade queue: a sphere or bounding
  // sphere with a material exists (a reflected ray may find it), or a material whose specular exponent is not
  // a small whole number
  bool lean_may_defer = false;
  // every member of the top-level list is a bounding sphere without a material (what loadObjFile builds):
  // no reflection or shadow ray can change a pixel
  bool inert_secondary = false;
  int exact_grid = 2048;  // workgroups of the grid-stride k_search_exact: eight per compute unit
  int fast_shade_grid = 1280;  // of k_shade_candidates_fast: as many as are resident at once (RTMI_SHADE_FAST_WAVES per SIMD)
  // one frame may be in flight between rtmi_render_begin and rtmi_render_end
  bool pending = false;
  DevCounters* h_cnt = nullptr;  // pinned host copy of the device counters
  std::vector<std::array<hipEvent_t, 4>> ev_pool;  // per search launch: before, after the search, after its shading, after the walk or scan kernel alone; reused
  size_t ev_used = 0;
  long long pend_launches = 0;
  int64_t pend_live_samples = 0;
  bool pend_shade_timed = false;
  bool pend_fast = false;  // the frame in flight runs without a ray-generation launch
  int64_t pend_work_bytes = 0;  // device memory of the work buffers the frame uses
  rtmi_camera pend_cam{};
  rtmi_params pend_params{};
  void* pend_rgba = nullptr;
  void* pend_hit = nullptr;
  std::mutex mu;
};

static int32_t validate_scene(const rtmi_scene_desc* s) {
  if (!s) return RTMI_ERR_INVALID_ARGUMENT;
  if (s->n_lights < 0 || s->n_materials < 0 || s->n_objects < 0 || s->n_triangles < 0) return RTMI_ERR_INVALID_ARGUMENT;
  if ((s->n_lights && !s->lights) || (s->n_materials && !s->materials) || (s->n_objects && !s->objects) ||
      (s->n_triangles && !s->triangles))
    return RTMI_ERR_INVALID_ARGUMENT;
  if (s->n_triangles > 0x7ffffff0LL || s->n_objects > 0x3fffffffLL) return RTMI_ERR_UNSUPPORTED;
  for (int64_t k = 0; k < s->n_objects; ++k) {
    const rtmi_object& ob = s->objects[k];
    if (ob.material >= s->n_materials) return RTMI_ERR_INVALID_ARGUMENT;
    switch (ob.kind) {
      case RTMI_SPHERE: break;
      case RTMI_TRIANGLES:
      case RTMI_BOUNDING_SPHERE:
        if (ob.first < 0 || ob.count < 0 || ob.first + ob.count > s->n_triangles) return RTMI_ERR_INVALID_ARGUMENT;
        break;
      case RTMI_LIGHT:
        if (ob.first < 0 || ob.first >= s->n_lights) return RTMI_ERR_INVALID_ARGUMENT;
        break;
      default: return RTMI_ERR_INVALID_ARGUMENT;
    }
  }
  for (int64_t i = 0; i < s->n_triangles; ++i)
    if (s->triangles[i].material >= s->n_materials) return RTMI_ERR_INVALID_ARGUMENT;
  return RTMI_OK;
}

static int32_t alloc_group(TriGroup& g, const std::vector<int>& idx, const rtmi_triangle* tris) {
  // An empty group (a bounding sphere without children) still gets its buffers: the search
  // kernel then runs with npacked == 0 and resolves every queued sample as a miss.
  g.n = (int)idx.size();
  const size_t n1 = std::max<size_t>(g.n, 1);
  const int nblocks = (int)((n1 + 255) / 256);
  HIP_TRY(hipMalloc(&g.d_group, sizeof(int) * n1));
  if (g.n) HIP_TRY(hipMemcpy(g.d_group, idx.data(), sizeof(int) * g.n, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&g.d_keep, n1));
  HIP_TRY(hipMalloc(&g.d_block_counts, sizeof(int) * nblocks));
  g.cone_bytes = sizeof(float) * CONE_DWORDS * ((size_t)(g.n + TRI_BATCH - 1) / TRI_BATCH + 2);
  HIP_TRY(hipMalloc(&g.d_cone, g.cone_bytes));
  HIP_TRY(hipMemset(g.d_cone, 0, g.cone_bytes));
  HIP_TRY(hipMalloc(&g.d_edge, sizeof(float) * EDGE_FLOATS * n1));
  HIP_TRY(hipMalloc(&g.d_packed_tri, sizeof(int) * n1));
  HIP_TRY(hipMalloc(&g.d_npacked, sizeof(int)));
  HIP_TRY(hipMemset(g.d_npacked, 0, sizeof(int)));

  TreeHost& th = g.tree_host;
  tree_build_host(tris, idx, &th);
  g.tree = th.dev;
  g.n_records = th.n_records;
  const size_t ns = th.slot_tri.size();
  HIP_TRY(hipMalloc(&g.d_slot_tri, sizeof(int) * ns));
  HIP_TRY(hipMemcpy(g.d_slot_tri, th.slot_tri.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&g.d_slot_pos, sizeof(int) * ns));
  HIP_TRY(hipMemcpy(g.d_slot_pos, th.slot_pos.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
  // entries of nodes that do not exist (the tail of a level's last record) stay zero for ever
  HIP_TRY(hipMalloc(&g.d_rec, sizeof(float) * REC_DWORDS * th.n_records));
  HIP_TRY(hipMemset(g.d_rec, 0, sizeof(float) * REC_DWORDS * th.n_records));
  HIP_TRY(hipMalloc(&g.d_tedge, sizeof(float) * EDGE_FLOATS * ns));
  HIP_TRY(hipMalloc(&g.d_nodes, sizeof(NodeCone) * th.n_node_buf));
  HIP_TRY(hipMalloc(&g.d_ticket, sizeof(unsigned int)));
  HIP_TRY(hipMemset(g.d_ticket, 0, sizeof(unsigned int)));
  HIP_TRY(hipMalloc(&g.d_pos_plane, sizeof(float4) * n1));
  HIP_TRY(hipMemset(g.d_pos_plane, 0, sizeof(float4) * n1));
  HIP_TRY(hipMalloc(&g.d_pos_mt, sizeof(int2) * n1));
  HIP_TRY(hipMemset(g.d_pos_mt, 0, sizeof(int2) * n1));
  g.tree.slot_tri = g.d_slot_tri;
  g.tree.slot_pos = g.d_slot_pos;
  g.tree.rec = g.d_rec;
  g.tree.edge = g.d_tedge;
  g.tree.nodes = g.d_nodes;
  g.tree.ticket = g.d_ticket;
  return RTMI_OK;
}

extern "C" int32_t rtmi_abi_version(void) { return RTMI_ABI_VERSION; }

extern "C" const char* rtmi_status_string(int32_t status) {
  switch (status) {
    case RTMI_OK: return "ok";
    case RTMI_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RTMI_ERR_UNSUPPORTED: return "unsupported input";
    case RTMI_ERR_NO_DEVICE: return "no usable gfx950 device";
    case RTMI_ERR_HIP: return "HIP runtime error";
    case RTMI_ERR_OUT_OF_MEMORY: return "out of device memory";
    default: return "unknown status";
  }
}

extern "C" const char* rtmi_last_error(void) { return g_last_error.c_str(); }

extern "C" int32_t rtmi_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return RTMI_ERR_NO_DEVICE;
  }
  return n;
}

extern "C" int32_t rtmi_samples_per_pixel(double s) {
  if (!(s > 0.0) || s > 1.0) return RTMI_ERR_INVALID_ARGUMENT;
  int n = 0;
  while (fma((double)n, s, 0.0) < 1.0) {
    if (++n > 64) return RTMI_ERR_UNSUPPORTED;
  }
  return n * n;
}

// Sphere.init(boundingObjects:), Renderables.swift:215-246 with Triangle.getBounds :169-195.
extern "C" int32_t rtmi_bounding_sphere(const rtmi_triangle* tris, int64_t n, double center[3], double* radius) {
  if ((!tris && n > 0) || n < 0 || !center || !radius) return RTMI_ERR_INVALID_ARGUMENT;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) {
        double c = tris[i].v[k][a];
        if (lo[a] > c) lo[a] = c;
        if (hi[a] < c) hi[a] = c;
      }
  double ext[3];
  for (int a = 0; a < 3; ++a) ext[a] = fabs(hi[a] - lo[a]);
  *radius = std::max(ext[0], std::max(ext[1], ext[2])) / 2;
  for (int a = 0; a < 3; ++a) center[a] = hi[a] - (ext[a] / 2);
  return RTMI_OK;
}

// The spatial slot order the cone tree gives a triangle list (host arithmetic only).
extern "C" int32_t rtmi_tree_slot_order(const rtmi_triangle* tris, int64_t n, int32_t* slot_pos, int64_t capacity,
                                        int64_t* n_slots) {
  if ((!tris && n > 0) || n < 0 || n > 0x7ffffff0LL || !n_slots) return RTMI_ERR_INVALID_ARGUMENT;
  return guarded([&]() -> int32_t {
    std::vector<int> idx((size_t)n);
    for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = (int)i;
    TreeHost th;
    tree_build_host(tris, idx, &th);
    *n_slots = (int64_t)th.slot_pos.size();
    if (slot_pos) {
      if (capacity < *n_slots) return RTMI_ERR_INVALID_ARGUMENT;
      std::copy(th.slot_pos.begin(), th.slot_pos.end(), slot_pos);
    }
    return RTMI_OK;
  });
}

extern "C" void rtmi_scene_destroy(rtmi_scene* sc) {
  if (!sc) return;
  (void)hipSetDevice(sc->device);
  if (sc->stream) (void)hipStreamSynchronize(sc->stream);
  for (auto& g : sc->groups) {
    (void)hipFree(g.d_group); (void)hipFree(g.d_keep); (void)hipFree(g.d_block_counts);
    (void)hipFree(g.d_cone); (void)hipFree(g.d_edge); (void)hipFree(g.d_packed_tri); (void)hipFree(g.d_npacked);
    (void)hipFree(g.d_slot_tri); (void)hipFree(g.d_slot_pos); (void)hipFree(g.d_rec); (void)hipFree(g.d_tedge);
    (void)hipFree(g.d_nodes); (void)hipFree(g.d_ticket); (void)hipFree(g.d_pos_plane); (void)hipFree(g.d_pos_mt);
  }
  (void)hipFree(sc->fast.hit_id); (void)hipFree(sc->fast.best_pos); (void)hipFree(sc->fast.rgbf); (void)hipFree(sc->fast.cand);
  (void)hipFree(sc->fast.multi); (void)hipFree(sc->wb.multi); (void)hipFree(sc->fast.live_blocks); (void)hipFree(sc->fast.block_slot);
  (void)hipFree(sc->fast.fix_list); (void)hipFree(sc->fast.consts); (void)hipFree(sc->fast.entries);
  (void)hipFree(sc->d_objects); (void)hipFree(sc->d_lights); (void)hipFree(sc->d_materials);
  (void)hipFree(sc->d_tris); (void)hipFree(sc->d_planes); (void)hipFree(sc->d_tri_owner);
  (void)hipFree(sc->d_trec); (void)hipFree(sc->d_gst2);
  (void)hipFree(sc->d_bmembers); (void)hipFree(sc->d_bowner); (void)hipFree(sc->d_bslot_pos);
  (void)hipFree(sc->d_brec); (void)hipFree(sc->d_bgst2);
  (void)hipFree(sc->wb.hit_id); (void)hipFree(sc->wb.tri_hit); (void)hipFree(sc->wb.q_child);
  (void)hipFree(sc->wb.q_shade); (void)hipFree(sc->wb.q_top); (void)hipFree(sc->wb.rgb);
  (void)hipFree(sc->wb.dirs); (void)hipFree(sc->wb.tval); (void)hipFree(sc->wb.best_pos); (void)hipFree(sc->wb.cand); (void)hipFree(sc->wb.tabs); (void)hipFree(sc->d_sargs); (void)hipFree(sc->d_genf);
  (void)hipFree(sc->wb.blocks); (void)hipFree(sc->wb.block_live); (void)hipFree(sc->wb.cnt);
  if (sc->h_cnt) (void)hipHostFree(sc->h_cnt);
  for (auto& tr : sc->ev_pool) for (hipEvent_t e : tr) (void)hipEventDestroy(e);
  if (sc->ev_a) (void)hipEventDestroy(sc->ev_a);
  if (sc->ev_b) (void)hipEventDestroy(sc->ev_b);
  if (sc->ev_c) (void)hipEventDestroy(sc->ev_c);
  if (sc->ev_d) (void)hipEventDestroy(sc->ev_d);
  if (sc->stream) (void)hipStreamDestroy(sc->stream);
  delete sc;
}

// The frame of reference of a general-origin filter over the triangles `members`: c0 = the centre of their
// bounding box (FP32 coordinates are taken relative to it) and r_o1, a bound on |o - c0|_1 over every point a
// secondary ray can start from -- all triangles, spheres and lights listed as objects.
static void general_frame(const rtmi_triangle* tris, int64_t n_tris, const rtmi_object* objects, int64_t n_objects,
                          const rtmi_light* lights, const std::vector<int>& members, double c0[3], double* r_o1_out) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  double slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
  auto grow = [](double* l, double* h, const double* p, double r) {
    // non-finite coordinates (such triangles are never hit) must not reach the FP32 frame of reference
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]) || !std::isfinite(r)) return;
    for (int a = 0; a < 3; ++a) { l[a] = std::min(l[a], p[a] - r); h[a] = std::max(h[a], p[a] + r); }
  };
  for (int ti : members) for (int k = 0; k < 3; ++k) grow(lo, hi, tris[ti].v[k], 0.0);
  // every surface a ray can start from: all triangles, spheres, lights listed as objects
  for (int64_t i = 0; i < n_tris; ++i) for (int k = 0; k < 3; ++k) grow(slo, shi, tris[i].v[k], 0.0);
  for (int64_t k = 0; k < n_objects; ++k) {
    const rtmi_object& ob = objects[k];
    if (ob.kind == RTMI_SPHERE || ob.kind == RTMI_BOUNDING_SPHERE) grow(slo, shi, ob.center, fabs(ob.radius));
    else if (ob.kind == RTMI_LIGHT) grow(slo, shi, lights[ob.first].loc, 0.0);
  }
  double r_o1 = 0.0;
  for (int a = 0; a < 3; ++a) {
    c0[a] = 0.5 * (lo[a] + hi[a]);
    r_o1 += std::max(fabs(slo[a] - c0[a]), fabs(shi[a] - c0[a]));
  }
  r_o1 *= 1.01;
  if (!std::isfinite(r_o1)) {  // no finite point at all: the filters then degrade to "always nominate" via their own checks
    r_o1 = 0.0;
    c0[0] = c0[1] = c0[2] = 0.0;
  }
  *r_o1_out = r_o1;
}

static int32_t scene_create_impl(const rtmi_scene_desc* desc, int32_t device, rtmi_scene* sc) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "no HIP device visible";
    return RTMI_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return RTMI_ERR_INVALID_ARGUMENT;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return RTMI_ERR_NO_DEVICE;
  }
  sc->device = device;
  {
    // workgroups of the grid-stride kernels: RTMI_GRID_PER_CU per compute unit (default 8), a multiple of NSLOT
    const char* e = getenv("RTMI_GRID_PER_CU");
    int per_cu = e ? atoi(e) : 8;
    if (per_cu < 1 || per_cu > 64) per_cu = 8;
    sc->exact_grid = ((std::max(1, prop.multiProcessorCount) * per_cu + NSLOT - 1) / NSLOT) * NSLOT;
    const char* e2 = getenv("RTMI_FAST_SHADE_GRID_PER_CU");
    int per_cu2 = e2 ? atoi(e2) : RTMI_SHADE_FAST_WAVES;
    if (per_cu2 < 1 || per_cu2 > 64) per_cu2 = RTMI_SHADE_FAST_WAVES;
    sc->fast_shade_grid = ((std::max(1, prop.multiProcessorCount) * per_cu2 + NSLOT - 1) / NSLOT) * NSLOT;
  }
  HIP_TRY(hipSetDevice(device));
  {
    int32_t rc0 = scratch_init(device);
    if (rc0 != RTMI_OK) return rc0;
  }
  HIP_TRY(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&sc->ev_a));
  HIP_TRY(hipEventCreate(&sc->ev_b));
  HIP_TRY(hipEventCreate(&sc->ev_c));
  HIP_TRY(hipEventCreate(&sc->ev_d));

  for (int32_t m = 0; m < desc->n_materials; ++m) {
    const double ns = desc->materials[m].Ns;
    if (!(ns >= 1.0 && ns <= 64.0 && ns == floor(ns))) sc->lean_may_defer = true;  // spec_pow_is_small_integer
  }
  sc->objects.assign(desc->objects, desc->objects + desc->n_objects);
  sc->h_lights.assign(desc->lights, desc->lights + desc->n_lights);
  sc->h_materials.assign(desc->materials, desc->materials + desc->n_materials);
  sc->n_lights = desc->n_lights;
  sc->n_materials = desc->n_materials;
  sc->n_tris = desc->n_triangles;

  auto upload = [&](void** dst, const void* src, size_t bytes) -> int32_t {
    *dst = nullptr;
    if (bytes == 0) return RTMI_OK;
    HIP_TRY(hipMalloc(dst, bytes));
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return RTMI_OK;
  };
  int32_t rc;
  if ((rc = upload((void**)&sc->d_objects, desc->objects, sizeof(rtmi_object) * desc->n_objects))) return rc;
  if ((rc = upload((void**)&sc->d_lights, desc->lights, sizeof(rtmi_light) * desc->n_lights))) return rc;
  if ((rc = upload((void**)&sc->d_materials, desc->materials, sizeof(rtmi_material) * desc->n_materials))) return rc;
  if ((rc = upload((void**)&sc->d_tris, desc->triangles, sizeof(rtmi_triangle) * desc->n_triangles))) return rc;

  // triangle -> owning object; groups
  std::vector<int> owner((size_t)desc->n_triangles, 0x7fffffff);
  std::vector<int> top;
  sc->groups.clear();
  sc->groups.emplace_back();
  sc->groups[0].object = -1;
  int n_sphere_like = 0;
  std::vector<std::vector<int>> child_lists;
  for (int64_t k = 0; k < desc->n_objects; ++k) {
    const rtmi_object& ob = desc->objects[k];
    if (ob.kind == RTMI_TRIANGLES) {
      for (int64_t i = ob.first; i < ob.first + ob.count; ++i) {
        top.push_back((int)i);
        // a triangle listed by several runs keeps its first (lowest) position
        owner[i] = std::min(owner[i], (int)k);
      }
    } else {
      ++n_sphere_like;
      if (ob.kind != RTMI_BOUNDING_SPHERE) ++sc->n_plain_shadeable;
      if (ob.kind != RTMI_LIGHT && ob.material >= 0) sc->lean_may_defer = true;
      if (ob.kind == RTMI_BOUNDING_SPHERE) {
        TriGroup g;
        g.object = (int)k;
        sc->groups.push_back(g);
        std::vector<int> idx;
        for (int64_t i = ob.first; i < ob.first + ob.count; ++i) idx.push_back((int)i);
        child_lists.push_back(std::move(idx));
      }
    }
  }
  sc->inert_secondary = desc->n_objects > 0;
  for (int64_t k = 0; k < desc->n_objects; ++k)
    if (desc->objects[k].kind != RTMI_BOUNDING_SPHERE || desc->objects[k].material >= 0) sc->inert_secondary = false;
  // a top-level triangle that appears in two runs would need per-position owners
  {
    std::vector<int> sorted = top;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return RTMI_ERR_UNSUPPORTED;
  }
  if ((rc = alloc_group(sc->groups[0], top, desc->triangles))) return rc;
  for (size_t g = 1; g < sc->groups.size(); ++g)
    if ((rc = alloc_group(sc->groups[g], child_lists[g - 1], desc->triangles))) return rc;
  if ((rc = upload((void**)&sc->d_tri_owner, owner.data(), sizeof(int) * owner.size()))) return rc;

  if (desc->n_triangles > 0) {
    HIP_TRY(hipMalloc(&sc->d_planes, sizeof(TriPlane) * desc->n_triangles));
    int nb = (int)((desc->n_triangles + 255) / 256);
    hipLaunchKernelGGL(k_tri_planes, dim3(nb), dim3(256), 0, sc->stream, sc->d_tris, (long long)desc->n_triangles, sc->d_planes);
    HIP_TRY(hipGetLastError());
  }
  // general-origin filter for secondary rays against the top-level triangle runs
  double c0[3] = {0.0, 0.0, 0.0};
  if (!top.empty()) {
    double r_o1 = 0.0;
    general_frame(desc->triangles, desc->n_triangles, desc->objects, desc->n_objects, desc->lights, top, c0, &r_o1);
    const size_t nt = top.size();
    HIP_TRY(hipMalloc(&sc->d_gst2, sizeof(float) * GST2_FLOATS * nt));
    hipLaunchKernelGGL(k_pack_general, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, sc->stream, sc->d_tris,
                       sc->groups[0].d_group, (int)nt, rtx::mk(c0[0], c0[1], c0[2]), r_o1, sc->d_gst2);
    HIP_TRY(hipGetLastError());
    // bounding-sphere tree over the slot order of group 0
    std::vector<float> trec;
    unsigned cone_levels = 0;
    tree_sphere_records(desc->triangles, sc->groups[0].tree_host, c0, r_o1, &trec, &cone_levels);
    sc->top_args.cone_levels = normal_cones_enabled() ? (int)cone_levels : 0;
    if ((rc = upload((void**)&sc->d_trec, trec.data(), sizeof(float) * trec.size()))) return rc;
    const TreeDev& td = sc->groups[0].tree;
    sc->top_args.ltop = td.ltop;
    sc->top_args.n_root = td.n_root;
    sc->top_args.n_slots = td.n_slots;
    sc->top_args.n_records = (int)sc->groups[0].n_records;
    for (int l = 0; l < TREE_MAX_LEVELS; ++l) sc->top_args.off[l] = td.rec_off[l];
  }
  for (auto& g : sc->groups) g.tree_host = TreeHost();  // host copies are not needed any more
  HIP_TRY(hipMalloc(&sc->wb.cnt, sizeof(DevCounters)));
  // one triple of timing events per triangle group: what a frame of one batch needs
  for (size_t g = 0; g < sc->groups.size(); ++g) {
    std::array<hipEvent_t, 4> t{};
    for (hipEvent_t& e : t) HIP_TRY(hipEventCreate(&e));
    sc->ev_pool.push_back(t);
  }
  HIP_TRY(hipMalloc(&sc->d_genf, sizeof(float4) * std::max(1, n_sphere_like)));
  HIP_TRY(hipHostMalloc((void**)&sc->h_cnt, sizeof(DevCounters), hipHostMallocDefault));
  sc->h_sargs.resize(sc->groups.size());
  HIP_TRY(hipMalloc(&sc->d_sargs, sizeof(SearchArgs) * sc->groups.size()));
  HIP_TRY(hipStreamSynchronize(sc->stream));

  sc->dev.objects = sc->d_objects;
  sc->dev.lights = sc->d_lights;
  sc->dev.materials = sc->d_materials;
  sc->dev.tris = sc->d_tris;
  sc->dev.planes = sc->d_planes;
  sc->dev.n_objects = (int)desc->n_objects;
  sc->dev.n_tris = (int)desc->n_triangles;
  sc->dev.n_lights = desc->n_lights;
  sc->dev.n_sphere_like = n_sphere_like;
  sc->dev.n_top_tris = (long long)top.size();
  sc->dev.top_group = sc->groups[0].d_group;
  sc->dev.tri_owner = sc->d_tri_owner;
  for (int a = 0; a < 3; ++a) sc->dev.c0[a] = c0[a];
  return RTMI_OK;
}

extern "C" int32_t rtmi_scene_create(const rtmi_scene_desc* desc, int32_t device, rtmi_scene** out) {
  if (!out) return RTMI_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int32_t rc = validate_scene(desc);
  if (rc != RTMI_OK) return rc;
  rtmi_scene* sc = new (std::nothrow) rtmi_scene();
  if (!sc) return RTMI_ERR_OUT_OF_MEMORY;
  rc = guarded([&] { return scene_create_impl(desc, device, sc); });
  if (rc != RTMI_OK) {
    rtmi_scene_destroy(sc);
    return rc;
  }
  *out = sc;
  return RTMI_OK;
}

static int32_t ensure_buffers(rtmi_scene* sc, size_t samples, size_t blocks) {
  if (sc->no_alloc && (samples > sc->wb.cap_samples || blocks > sc->wb.cap_blocks)) {
    g_last_error = "rtmi_render_device_noalloc: the frame needs " + std::to_string(samples) + " samples of work buffers in " +
                   std::to_string(blocks) + " pixel blocks; the static array holds " + std::to_string(sc->wb.cap_samples) +
                   " samples in " + std::to_string(sc->wb.cap_blocks) + " blocks";
    return RTMI_ERR_UNSUPPORTED;
  }
  if (samples > sc->wb.cap_samples) {
    (void)hipFree(sc->wb.hit_id); (void)hipFree(sc->wb.tri_hit); (void)hipFree(sc->wb.q_child);
    (void)hipFree(sc->wb.q_shade); (void)hipFree(sc->wb.q_top); (void)hipFree(sc->wb.rgb);
    (void)hipFree(sc->wb.dirs); (void)hipFree(sc->wb.tval); (void)hipFree(sc->wb.best_pos); (void)hipFree(sc->wb.cand);
    (void)hipFree(sc->wb.multi);
    sc->wb.multi = nullptr;
    sc->wb.cap_multi = 0;
    sc->wb.hit_id = sc->wb.tri_hit = sc->wb.q_child = sc->wb.q_shade = sc->wb.q_top = sc->wb.best_pos = nullptr;
    sc->wb.rgb = sc->wb.dirs = sc->wb.tval = nullptr;
    sc->wb.cand = nullptr;
    sc->wb.cap_samples = 0;
    HIP_TRY(hipMalloc(&sc->wb.hit_id, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->wb.tri_hit, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->wb.q_child, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->wb.q_shade, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->wb.q_top, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->wb.rgb, sizeof(double) * 3 * samples));
    HIP_TRY(hipMalloc(&sc->wb.dirs, sizeof(double) * 3 * samples));
    HIP_TRY(hipMalloc(&sc->wb.tval, sizeof(double) * samples));
    HIP_TRY(hipMalloc(&sc->wb.best_pos, sizeof(int) * samples));
    // A closed mesh that fills the frame yields about one candidate per sample; twice that, plus a
    // floor that keeps each of the NSLOT lists roomy when few waves share few lists (small frames).
    sc->wb.cap_cand = 2 * samples + (size_t)NSLOT * 16384;
    // RTMI_CAND_CAPACITY (entries, for tests): a small value makes the lists overflow, which sends the frame
    // through the list scan instead (render_end)
    if (const char* e = getenv("RTMI_CAND_CAPACITY")) {
      const long long v = atoll(e);
      if (v > 0) sc->wb.cap_cand = (size_t)std::max<long long>(v, NSLOT);
    }
    HIP_TRY(hipMalloc(&sc->wb.cand, sizeof(int2) * sc->wb.cap_cand));
    // a closed mesh seen from outside gives a few rays in a hundred a second candidate; if this list does not fit,
    // the decisions scan the candidate lists instead
    sc->wb.cap_multi = std::min(sc->wb.cap_cand, samples / 2 + (size_t)NSLOT * 4096);
    if (const char* e = getenv("RTMI_MULTI_CAPACITY")) {  // tests: a small value makes this list overflow
      const long long v = atoll(e);
      if (v > 0) sc->wb.cap_multi = (size_t)std::max<long long>(v, NSLOT);
    }
    HIP_TRY(hipMalloc(&sc->wb.multi, sizeof(int2) * sc->wb.cap_multi));
    sc->wb.cap_samples = samples;
  }
  if (blocks > sc->wb.cap_blocks) {
    (void)hipFree(sc->wb.blocks);
    (void)hipFree(sc->wb.block_live);
    sc->wb.blocks = nullptr;
    sc->wb.block_live = nullptr;
    sc->wb.cap_blocks = 0;
    sc->blocks_valid = false;
    HIP_TRY(hipMalloc(&sc->wb.blocks, sizeof(int2) * blocks));
    HIP_TRY(hipMalloc(&sc->wb.block_live, blocks));
    sc->wb.cap_blocks = blocks;
  }
  return RTMI_OK;
}

// Per-frame cone tree of one group for the frame's camera position.
static int32_t pack_group_tree(rtmi_scene* sc, TriGroup& g, d3 cam, double dir_err) {
  if (g.n == 0) return RTMI_OK;  // the records were zeroed at creation: nothing passes
  int nblocks = (g.tree.n_slots + TREE_BLOCK - 1) / TREE_BLOCK;
  hipLaunchKernelGGL(k_pack_tree, dim3(nblocks), dim3(TREE_BLOCK), 0, sc->stream, sc->d_tris, sc->d_planes, g.tree, cam, dir_err,
                     g.d_pos_plane, g.d_pos_mt);
  HIP_TRY(hipGetLastError());
  return RTMI_OK;
}

// Run the list-scan pack for one group against the frame's camera.
static int32_t pack_group(rtmi_scene* sc, TriGroup& g, d3 cam) {
  if (g.n == 0) return RTMI_OK;
  int nblocks = (g.n + 255) / 256;
  HIP_TRY(hipMemsetAsync(g.d_cone, 0, g.cone_bytes, sc->stream));  // zero cone vector: never nominates; flags clear
  hipLaunchKernelGGL(k_pack_flags, dim3(nblocks), dim3(256), 0, sc->stream, sc->d_tris, sc->d_planes, g.d_group, g.n,
                     cam, g.d_keep, g.d_block_counts);
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, sc->stream, g.d_block_counts, nblocks, g.d_npacked);
  hipLaunchKernelGGL(k_pack_write, dim3(nblocks), dim3(256), 0, sc->stream, sc->d_tris, g.d_group, g.n, cam, g.d_keep,
                     g.d_block_counts, g.d_cone, g.d_edge, g.d_packed_tri);
  HIP_TRY(hipGetLastError());
  return RTMI_OK;
}

// Rays per lane of the search kernels; RTMI_SEARCH_R=2|4|8 overrides the choice for experiments.  The list scan
// takes 4.  The tree search takes 4 on large batches and 2 on small ones: a wave's walk is one serial chain of
// record fetches whatever the number of rays it carries, and a small batch (a 1080p frame at 4 samples per pixel is
// 8 M samples) has too few waves at 4 rays per lane to hide that chain -- measured on the 1.3 M-triangle mesh at
// 1080p: 1.40 ms per frame with 2 rays per lane, 1.67 with 4; at 4K and 16 samples per pixel 4 wins by 11 %.
// RTMI_FLAG_SHADOWS_ENTER_BOUNDED, RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED: the bounding-sphere tree over the children of all bounding spheres (the same
// construction as the top-level runs' tree in scene_create_impl), built when a frame first asks for the flag.
// The triangles come back from the device: the caller's arrays are gone by now.  The scene's stream is idle.
static int32_t ensure_bounded_tree(rtmi_scene* sc) {
  if (sc->bounded_tree_built) return RTMI_OK;
  std::vector<int> members, owner;
  for (size_t k = 0; k < sc->objects.size(); ++k) {
    const rtmi_object& ob = sc->objects[k];
    if (ob.kind != RTMI_BOUNDING_SPHERE) continue;
    if (members.size() + (size_t)ob.count > 0x7ffffff0ull) return RTMI_ERR_UNSUPPORTED;
    for (int64_t i = ob.first; i < ob.first + ob.count; ++i) {
      members.push_back((int)i);
      owner.push_back((int)k);
    }
  }
  ExtArgs args{};
  if (!members.empty()) {
    // the five device arrays are built into locals and handed to the scene only when all of them exist: a failure
    // half-way (out of memory) frees what it made and leaves the scene as it was
    void* made[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // members, owner, slot_pos, gst2, records
    auto build = [&]() -> int32_t {
      HIP_TRY(hipStreamSynchronize(sc->stream));
      std::vector<rtmi_triangle> tris((size_t)sc->n_tris);
      HIP_TRY(hipMemcpy(tris.data(), sc->d_tris, sizeof(rtmi_triangle) * tris.size(), hipMemcpyDeviceToHost));
      double c0[3] = {0.0, 0.0, 0.0}, r_o1 = 0.0;
      general_frame(tris.data(), sc->n_tris, sc->objects.data(), (int64_t)sc->objects.size(), sc->h_lights.data(), members, c0, &r_o1);
      TreeHost th;
      tree_build_host(tris.data(), members, &th);
      auto upload = [&](void** dst, const void* src, size_t bytes) -> int32_t {
        HIP_TRY(hipMalloc(dst, bytes));
        HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return RTMI_OK;
      };
      int32_t rc;
      const size_t nm = members.size();
      if ((rc = upload(&made[0], members.data(), sizeof(int) * nm))) return rc;
      if ((rc = upload(&made[1], owner.data(), sizeof(int) * nm))) return rc;
      if ((rc = upload(&made[2], th.slot_pos.data(), sizeof(int) * th.slot_pos.size()))) return rc;
      HIP_TRY(hipMalloc(&made[3], sizeof(float) * GST2_FLOATS * nm));
      hipLaunchKernelGGL(k_pack_general, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, sc->stream, sc->d_tris,
                         (const int*)made[0], (int)nm, rtx::mk(c0[0], c0[1], c0[2]), r_o1, (float*)made[3]);
      HIP_TRY(hipGetLastError());
      std::vector<float> trec;
      unsigned cone_levels = 0;
      tree_sphere_records(tris.data(), th, c0, r_o1, &trec, &cone_levels);
      if ((rc = upload(&made[4], trec.data(), sizeof(float) * trec.size()))) return rc;
      args.tree.cone_levels = normal_cones_enabled() ? (int)cone_levels : 0;
      args.tree.ltop = th.dev.ltop;
      args.tree.n_root = th.dev.n_root;
      args.tree.n_slots = th.dev.n_slots;
      args.tree.n_records = (int)th.n_records;
      for (int l = 0; l < TREE_MAX_LEVELS; ++l) args.tree.off[l] = th.dev.rec_off[l];
      args.n_bounded = (long long)nm;
      for (int a = 0; a < 3; ++a) args.c0[a] = c0[a];
      HIP_TRY(hipStreamSynchronize(sc->stream));
      return RTMI_OK;
    };
    const int32_t rc = guarded(build);
    if (rc != RTMI_OK) {
      (void)hipStreamSynchronize(sc->stream);  // k_pack_general may still be writing made[3]
      for (void* p : made) (void)hipFree(p);
      return rc;
    }
    sc->d_bmembers = (int*)made[0];
    sc->d_bowner = (int*)made[1];
    sc->d_bslot_pos = (int*)made[2];
    sc->d_bgst2 = (float*)made[3];
    sc->d_brec = (float*)made[4];
  }
  sc->bounded_args = args;
  sc->bounded_tree_built = true;
  return RTMI_OK;
}

// `compacted`: the launch walks the live blocks only (k_search_tree_fast); its waves are all busy, and four rays per
// lane are the better choice down to an eighth of a 4K frame (measured: 0.27 against 0.32 ms) and for a 1024 x 1024 one
// -- unless the mesh is so fine that neighbouring rays meet different triangles (dense: 1.3 M triangles at 1080p with 4
// samples per pixel, 1.50 ms with two rays per lane, 1.79 with four).
static int search_rays_per_lane(bool tree, int nsamples, bool compacted = false, bool dense = false) {
  static int forced = [] {
    const char* e = getenv("RTMI_SEARCH_R");
    int v = e ? atoi(e) : 0;
    return (v == 8 || v == 2 || v == 4) ? v : 0;
  }();
  if (forced) return forced;
  if (compacted) return dense ? 2 : 4;
  return tree && nsamples <= (16 << 20) ? 2 : 4;
}
// Rays with a single candidate are decided where they are shaded (the default), or by k_search_exact like the others
// (RTMI_SINGLES=0, for comparison).
static bool singles_enabled() {
  static bool v = [] {
    const char* e = getenv("RTMI_SINGLES");
    return !(e && atoi(e) == 0);
  }();
  return v;
}
// Scenes with top-level triangle runs: trace the queued hits' secondary rays in one kernel and shade from the
// recorded outcomes in another (the default), or do both in one kernel (RTMI_TRACE_THEN_SHADE=0, for comparison).
static bool trace_then_shade() {
  static bool v = [] {
    const char* e = getenv("RTMI_TRACE_THEN_SHADE");
    return !(e && atoi(e) == 0);
  }();
  return v;
}
constexpr int MAX_FRAME_EDGE = 1 << 20;          // widest and tallest frame accepted, in pixels
constexpr int64_t MAX_FRAME_BLOCKS = 1ll << 26;  // 8x8 pixel blocks of a frame (the host keeps a list of them)
constexpr size_t MAX_BATCH_SAMPLES = 1u << 28;  // 256 Mi samples per batch (about 100 B of work buffers per sample)

// Frames with more samples than this are rendered in several batches of whole pixel blocks.
// RTMI_MAX_BATCH_SAMPLES lowers it (tests use that to take the multi-batch path with small frames).
static size_t max_batch_samples() {
  static size_t v = [] {
    const char* e = getenv("RTMI_MAX_BATCH_SAMPLES");
    long long x = e ? atoll(e) : 0;
    return x > 0 ? std::min<size_t>((size_t)x, MAX_BATCH_SAMPLES) : MAX_BATCH_SAMPLES;
  }();
  return v;
}

// The camera frame: RayTracingFunctions.swift:156-170 (focal-plane centre, the fixed 10 x 10 window, its corners) and
// WorldCoordinateSequence.swift:58-59 (pixels per world unit from the frame's last pixel index).
static void frame_basis(const rtmi_camera* cam, int width, int height, FrameConst* fc) {
  d3 camera = rtx::ld3(cam->pos);
  d3 center = rtx::add(camera, rtx::smul(cam->focal_length, rtx::ld3(cam->dir)));
  d3 u, v;
  rtx::plane_vectors(rtx::ld3(cam->dir), &u, &v);
  d3 hpc = rtx::muls(u, 10.0 / 2.0), vpc = rtx::muls(v, 10.0 / 2.0);
  d3 ul = rtx::add(rtx::sub(center, hpc), vpc);
  d3 ur = rtx::add(rtx::add(center, hpc), vpc);
  d3 ll = rtx::sub(rtx::sub(center, hpc), vpc);
  fc->ul[0] = ul.x; fc->ul[1] = ul.y; fc->ul[2] = ul.z;
  fc->u[0] = u.x; fc->u[1] = u.y; fc->u[2] = u.z;
  fc->v[0] = v.x; fc->v[1] = v.y; fc->v[2] = v.z;
  fc->cam[0] = camera.x; fc->cam[1] = camera.y; fc->cam[2] = camera.z;
  fc->ppw_h = (double)(width - 1) / sqrt(rtx::len_squared(rtx::sub(ur, ul)));
  fc->ppw_v = (double)(height - 1) / sqrt(rtx::len_squared(rtx::sub(ll, ul)));
  fc->W = width;
  fc->H = height;
}

// RTMI_FAST=0: frames that qualify for the path without a ray-generation launch take the general path (for comparison).
// The small uploads of a frame -- constants that change with the camera, the search argument blocks, the cleared
// counters -- travel as ONE kernel launch whose argument carries the bytes (a few hundred for a loadObjFile scene):
// separate hipMemsetAsync / hipMemcpyAsync calls cost about 10 us of stream time each, which is what a frame's share
// of an 8-way tile split can least afford.  Uploads that do not fit the argument go the usual way.
constexpr int UPLOAD_MAX_SEGS = 6;
constexpr int UPLOAD_BLOB_WORDS = 832;
struct UploadBlob {
  int n_seg;
  int zero_words;
  unsigned int* zero_dst;
  unsigned int* dst[UPLOAD_MAX_SEGS];
  int off_words[UPLOAD_MAX_SEGS];
  int words[UPLOAD_MAX_SEGS];
  unsigned int data[UPLOAD_BLOB_WORDS];
};
static_assert(sizeof(UploadBlob) <= 3584, "a kernel argument block holds 4 KB");
__global__ void __launch_bounds__(256) k_frame_uploads(UploadBlob b) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
  for (int i = tid; i < b.zero_words; i += nthreads) b.zero_dst[i] = 0u;
  for (int sgm = 0; sgm < b.n_seg; ++sgm)
    for (int i = tid; i < b.words[sgm]; i += nthreads) b.dst[sgm][i] = b.data[b.off_words[sgm] + i];
}
struct FrameUploads {
  struct Seg { void* dst; const void* src; size_t bytes; };
  std::vector<Seg> segs;
  void* zero_dst = nullptr;
  size_t zero_bytes = 0;
  // src must stay untouched until the next upload to dst (the fallback's asynchronous copy reads it after the call returns)
  void add(void* dst, const void* src, size_t bytes) { if (bytes) segs.push_back({dst, src, bytes}); }
  void zero(void* dst, size_t bytes) { zero_dst = dst; zero_bytes = bytes; }
  int32_t flush(hipStream_t st) {
    if (segs.empty() && !zero_bytes) return RTMI_OK;
    size_t total = 0;
    bool words = zero_bytes % 4 == 0 && zero_bytes / 4 <= 0x7fffffff;
    for (const Seg& g : segs) { total += g.bytes; words = words && g.bytes % 4 == 0; }
    static const bool by_kernel = [] {  // RTMI_UPLOAD_KERNEL=0 (tests, comparisons): always the separate calls
      const char* e = getenv("RTMI_UPLOAD_KERNEL");
      return !(e && atoi(e) == 0);
    }();
    if (by_kernel && words && segs.size() <= (size_t)UPLOAD_MAX_SEGS && total <= sizeof(unsigned int) * UPLOAD_BLOB_WORDS) {
      UploadBlob b{};
      b.n_seg = (int)segs.size();
      b.zero_words = (int)(zero_bytes / 4);
      b.zero_dst = static_cast<unsigned int*>(zero_dst);
      int at = 0;
      for (size_t i = 0; i < segs.size(); ++i) {
        b.dst[i] = static_cast<unsigned int*>(segs[i].dst);
        b.off_words[i] = at;
        b.words[i] = (int)(segs[i].bytes / 4);
        memcpy(b.data + at, segs[i].src, segs[i].bytes);
        at += b.words[i];
      }
      hipLaunchKernelGGL(k_frame_uploads, dim3(8), dim3(256), 0, st, b);
      HIP_TRY(hipGetLastError());
    } else {
      if (zero_bytes) HIP_TRY(hipMemsetAsync(zero_dst, 0, zero_bytes, st));
      for (const Seg& g : segs) HIP_TRY(hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyHostToDevice, st));
    }
    segs.clear();
    zero_bytes = 0;
    return RTMI_OK;
  }
};

static bool fast_enabled() {
  static bool v = [] {
    const char* e = getenv("RTMI_FAST");
    return !(e && atoi(e) == 0);
  }();
  return v;
}
// RTMI_BLOCK_ENTRIES=0 (experiments): the walks of that path start at the root instead of the records found per block.
static bool block_entries_enabled() {
  static bool v = [] {
    const char* e = getenv("RTMI_BLOCK_ENTRIES");
    return !(e && atoi(e) == 0);
  }();
  return v;
}
// RTMI_LEAF_MASKS=0 (experiments): those start records name whole leaf records, never the triangles in them;
// 2 (tests): always the triangles, also where the mesh is so fine for the frame that it is not worth it.
static int leaf_masks_mode() {
  static int v = [] {
    const char* e = getenv("RTMI_LEAF_MASKS");
    return e ? atoi(e) : 1;
  }();
  return v;
}
// RTMI_FAST_FORCE (tests): bit 0 = every (ray, triangle) pair of such a frame is decided in FP64, bit 1 = every pixel is
// shaded again in FP64 -- the two fallbacks of the path, each made the rule.
static int fast_force() {
  static int v = [] {
    const char* e = getenv("RTMI_FAST_FORCE");
    return e ? atoi(e) & 3 : 0;
  }();
  return v;
}
constexpr int FAST_MAX_OBJECTS = 8;
constexpr int FAST_MAX_MATERIALS = 4096;

// Bound on |d - d*| for the FP32 sample directions of fast_dir, d* being the reference's FP64 unit direction
// (normalize(pixelToWorldCoordinate(...) - camera), WorldCoordinateSequence.swift:62-66, RayTracingFunctions.swift:73).
// fast_dir evaluates f = (base + uComponent) - vComponent with base = float(ul - camera) and the two components
// rounded to float from the FP64 tables: per axis k the three roundings and two additions err by at most
// 3 * 2^-24 * B_k, B_k = |base_k| + max|uComponent_k| + max|vComponent_k|; the reference's own FP64 roundings (it adds
// ul first and subtracts the camera last) differ from the exact value by a few 2^-53 of |ul|_1 + |camera|_1 + B.  A vector
// that is e away from f* normalises to a unit vector at most 2 e / |f*| away from d*, and |f*| is at least the distance
// dmin from the camera to the focal plane; the FP32 normalisation itself (three products, two sums, v_rsq_f32 at 1 ulp,
// three products) stays within 8 * 2^-24.  The result carries a factor 1.5 on top; tests/test_fast_path_math.py
// compares it with observed errors.  Returns +inf when the frame has no usable bound.
static double fast_dir_err(const FrameConst& fc, const rtmi_camera* cam) {
  const double u32 = 5.9604644775390625e-08;
  const double dlen = sqrt(cam->dir[0] * cam->dir[0] + cam->dir[1] * cam->dir[1] + cam->dir[2] * cam->dir[2]);
  const double dmin = fabs(cam->focal_length) * dlen;
  double b2 = 0.0, big = 0.0;
  for (int k = 0; k < 3; ++k) {
    const double bk = fabs(fc.ul[k] - fc.cam[k]) + fabs(fc.u[k]) * ((double)fc.W / fabs(fc.ppw_h)) + fabs(fc.v[k]) * ((double)fc.H / fabs(fc.ppw_v));
    b2 += bk * bk;
    big += fabs(fc.ul[k]) + fabs(fc.cam[k]) + bk;
  }
  const double ef = 3.0 * u32 * sqrt(b2) + 1.0e-15 * big;
  const double err = 1.5 * (2.0 * ef / dmin + 8.0 * u32);
  return std::isfinite(err) && dmin > 0 ? err : INFINITY;
}

// live_blocks: the 8x8 blocks (of one batch) the per-sample arrays must have room for; all_blocks: blocks of the largest batch.
static int32_t ensure_fast_buffers(rtmi_scene* sc, size_t live_blocks, size_t samples_per_block, size_t all_blocks, size_t consts_bytes) {
  FastBufs& f = sc->fast;
  const size_t samples = live_blocks * samples_per_block;
  if (samples > f.cap_samples || live_blocks > f.cap_live) {
    (void)hipFree(f.hit_id); (void)hipFree(f.best_pos); (void)hipFree(f.rgbf); (void)hipFree(f.cand); (void)hipFree(f.multi);
    (void)hipFree(f.live_blocks); (void)hipFree(f.entries);
    f.live_blocks = nullptr;
    f.entries = nullptr;
    f.cap_live = 0;
    f.multi = nullptr;
    f.cap_multi = 0;
    f.hit_id = f.best_pos = nullptr;
    f.rgbf = nullptr;
    f.cand = nullptr;
    f.cap_samples = 0;
    HIP_TRY(hipMalloc(&f.hit_id, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&f.best_pos, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&f.rgbf, sizeof(float4) * samples));
    f.cap_cand = 2 * samples + (size_t)NSLOT * 16384;  // as ensure_buffers
    if (const char* e = getenv("RTMI_CAND_CAPACITY")) {
      const long long v = atoll(e);
      if (v > 0) f.cap_cand = (size_t)std::max<long long>(v, NSLOT);
    }
    HIP_TRY(hipMalloc(&f.cand, sizeof(int2) * f.cap_cand));
    f.cap_multi = std::min(f.cap_cand, samples / 2 + (size_t)NSLOT * 4096);
    if (const char* e = getenv("RTMI_MULTI_CAPACITY")) {  // tests: a small value makes this list overflow
      const long long v = atoll(e);
      if (v > 0) f.cap_multi = (size_t)std::max<long long>(v, NSLOT);
    }
    HIP_TRY(hipMalloc(&f.multi, sizeof(int2) * f.cap_multi));
    HIP_TRY(hipMalloc(&f.live_blocks, sizeof(int2) * live_blocks));
    HIP_TRY(hipMalloc(&f.entries, sizeof(int) * ENTRY_K * live_blocks));
    f.cap_live = live_blocks;
    f.cap_samples = samples;
  }
  if (all_blocks > f.cap_slots) {
    (void)hipFree(f.block_slot);
    f.block_slot = nullptr;
    f.cap_slots = 0;
    HIP_TRY(hipMalloc(&f.block_slot, sizeof(int) * all_blocks));
    f.cap_slots = all_blocks;
  }
  // a list of k_resolve_fast holds every pixel of the workgroups that append to it
  const size_t pixels = all_blocks * BLK_PIX;
  const size_t fix = ((pixels + 255) / 256 + NSLOT - 1) / NSLOT * 256 * NSLOT;
  if (fix > f.cap_fix) {
    (void)hipFree(f.fix_list);
    f.fix_list = nullptr;
    f.cap_fix = 0;
    HIP_TRY(hipMalloc(&f.fix_list, sizeof(int) * fix));
    f.cap_fix = fix;
  }
  if (consts_bytes > f.cap_consts) {
    (void)hipFree(f.consts);
    f.consts = nullptr;
    f.cap_consts = 0;
    f.consts_uploaded.clear();
    HIP_TRY(hipMalloc(&f.consts, consts_bytes));
    f.cap_consts = consts_bytes;
  }
  return RTMI_OK;
}

// Blocks the live list of a batch of `batch_blocks` gets room for.  Half of them to start with (a mesh wrapped in its
// bounding sphere rarely fills the frame), all of them for small frames; a frame that needs more is rendered again with
// what it needed and a quarter on top (render_finish), and the scene keeps that.  RTMI_LIVE_CAPACITY (tests) sets the start.
static size_t fast_live_capacity(const rtmi_scene* sc, size_t batch_blocks) {
  size_t want = batch_blocks <= 4096 ? batch_blocks : std::max<size_t>(4096, batch_blocks / 2);
  if (const char* e = getenv("RTMI_LIVE_CAPACITY")) {
    const long long v = atoll(e);
    if (v > 0) want = (size_t)v;
  }
  want = std::max(want, sc->fast.live_wanted);
  return std::max<size_t>(1, std::min(want, batch_blocks));
}

// Enqueue a whole frame on the scene's stream; nothing here waits for the device.  The caller
// holds sc->mu.
static int32_t render_enqueue(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                              void* hit_dev) {
  if (!sc || !cam || !p || !rgba_dev) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width < 2 || p->height < 2) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width > MAX_FRAME_EDGE || p->height > MAX_FRAME_EDGE) return RTMI_ERR_UNSUPPORTED;  // index arithmetic below is sized for this
  if (((int64_t)p->width / BLK + 1) * ((int64_t)p->height / BLK + 1) > MAX_FRAME_BLOCKS) {
    g_last_error = "frames of more than 2^26 pixel blocks (about 4.3 gigapixels) are not supported";
    return RTMI_ERR_UNSUPPORTED;
  }
  if (p->hit_policy != RTMI_HIT_LIST_ORDER && p->hit_policy != RTMI_HIT_NEAREST) return RTMI_ERR_UNSUPPORTED;
  if (p->search != RTMI_SEARCH_CONE_TREE && p->search != RTMI_SEARCH_LIST_SCAN) return RTMI_ERR_INVALID_ARGUMENT;
  constexpr int EXT_FLAGS = RTMI_FLAG_SHADOW_MAX_DISTANCE | RTMI_FLAG_SHADOWS_ENTER_BOUNDED | RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED;
  constexpr int ENTER_FLAGS = RTMI_FLAG_SHADOWS_ENTER_BOUNDED | RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED;
  if (p->flags & ~(RTMI_FLAG_NO_RAY_STATISTICS | EXT_FLAGS)) return RTMI_ERR_INVALID_ARGUMENT;
  // the secondary-ray extensions: every hit is shaded from the shade queue
  int ext_flags = p->flags & EXT_FLAGS;
  // the distance limit alone, in a scene whose list holds nothing that can block a light (bounding spheres only, what
  // loadObjFile builds): the frame is the reference's
  if (ext_flags == RTMI_FLAG_SHADOW_MAX_DISTANCE && sc->groups[0].n == 0 && sc->n_plain_shadeable == 0) ext_flags = 0;
  const bool ext = ext_flags != 0;
  if (ext && sc->no_alloc) {
    g_last_error = "rtmi_render_device_noalloc: the secondary-ray extension flags are not supported";
    return RTMI_ERR_UNSUPPORTED;
  }
  const bool use_tree = p->search == RTMI_SEARCH_CONE_TREE;
  const bool nearest = p->hit_policy == RTMI_HIT_NEAREST;
  if (nearest && !use_tree) {
    g_last_error = "RTMI_HIT_NEAREST needs the cone tree search";
    return RTMI_ERR_UNSUPPORTED;
  }
  if (nearest)
    for (const TriGroup& g : sc->groups)
      if (g.n >= NEAR_MARK) return RTMI_ERR_UNSUPPORTED;  // list positions share a word with the mark of a hit
  int spp = rtmi_samples_per_pixel(p->pixel_subdivision);
  if (spp < 0) return spp;
  int n1 = 1;
  while (n1 * n1 < spp) ++n1;
  int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
  if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = p->width - 1; y1 = p->height - 1; }
  if (x0 < 0 || y0 < 0 || x1 > p->width - 1 || y1 > p->height - 1 || x0 > x1 || y0 > y1) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->tile_size < 0 || p->tile_world < 0 || (p->tile_world > 1 && (p->tile_rank < 0 || p->tile_rank >= p->tile_world)))
    return RTMI_ERR_INVALID_ARGUMENT;
  if (p->tile_world > 1 && p->tile_size <= 0) return RTMI_ERR_INVALID_ARGUMENT;
  // stride(from: x, to: x + 1, by: s) yields the values fma(k, s, x) < x + 1 (WorldCoordinateSequence.swift:74-75).
  // The kernels assume the same count n1 for every pixel of the window; for a subdivision that puts
  // some k*s within a rounding error of 1 the count can depend on x, and such a frame is refused.
  {
    const double s = p->pixel_subdivision;
    auto uniform = [&](int lo, int hi) {
      for (int c = lo; c < hi; ++c) {
        const double x = (double)c;
        if (!(fma((double)(n1 - 1), s, x) < x + 1.0) || fma((double)n1, s, x) < x + 1.0) return false;
      }
      return true;
    };
    if (!uniform(x0, x1) || !uniform(y0, y1)) {
      g_last_error = "pixel_subdivision gives a sample count that varies from pixel to pixel";
      return RTMI_ERR_UNSUPPORTED;
    }
  }

  if (sc->pending) {
    g_last_error = "a frame is already in flight on this scene; call rtmi_render_end first";
    return RTMI_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(sc->device));
  ExtArgs ext_args{};
  if (ext_flags & ENTER_FLAGS) {
    int32_t rc_b = ensure_bounded_tree(sc);
    if (rc_b) return rc_b;
    ext_args = sc->bounded_args;
  }
  ext_args.flags = ext_flags;

  // camera frame: RayTracingFunctions.swift:156-170, WorldCoordinateSequence.swift:58-59
  FrameConst fc{};
  frame_basis(cam, p->width, p->height, &fc);
  fc.nearest = nearest ? 1 : 0;
  fc.sub = p->pixel_subdivision;
  for (int a = 0; a < 3; ++a) fc.ambient[a] = p->ambient[a];
  fc.W = p->width; fc.H = p->height;
  fc.x0 = x0; fc.y0 = y0; fc.x1 = x1; fc.y1 = y1;
  fc.spp = spp; fc.n1 = n1;
  fc.spp_shift = fc.n1_shift = -1;
  for (int sh = 0; sh < 16; ++sh) {
    if ((1 << sh) == spp) fc.spp_shift = sh;
    if ((1 << sh) == n1) fc.n1_shift = sh;
  }
  fc.tile_size = p->tile_size;
  fc.tiles_x = p->tile_size > 0 ? (p->width - 1 + p->tile_size - 1) / p->tile_size : 1;
  fc.tile_rank = p->tile_rank;
  fc.tile_world = p->tile_world > 1 ? p->tile_world : 1;
  fc.inv_spp = 1.0f / (float)spp;
  {
    size_t need = (size_t)(p->width + p->height) * n1 * 3;
    if (need > sc->wb.cap_tabs && sc->no_alloc) {
      g_last_error = "rtmi_render_device_noalloc: (width + height) x sub-samples per axis = " + std::to_string(need / 3) +
                     " table entries; the static array holds " + std::to_string(sc->wb.cap_tabs / 3);
      return RTMI_ERR_UNSUPPORTED;
    }
    if (need > sc->wb.cap_tabs) {
      (void)hipFree(sc->wb.tabs);
      sc->wb.tabs = nullptr;
      sc->wb.cap_tabs = 0;
      HIP_TRY(hipMalloc(&sc->wb.tabs, (sizeof(double) + sizeof(float)) * need));  // FP64 tables, then their FP32 copies
      sc->wb.cap_tabs = need;
      sc->tabs_valid = false;
    }
    fc.utab = sc->wb.tabs;
    fc.vtab = sc->wb.tabs + (size_t)p->width * n1 * 3;
    fc.utab_f = reinterpret_cast<const float*>(sc->wb.tabs + sc->wb.cap_tabs);
    fc.vtab_f = fc.utab_f + (size_t)p->width * n1 * 3;
  }

  // 8x8 pixel blocks that contain at least one pixel to render.  The list depends only on the
  // frame geometry and tile selection, so it is rebuilt (and re-uploaded) only when those change.
  const std::array<int, 9> blocks_key = {p->width, p->height, x0, y0, x1, y1, fc.tile_size, fc.tile_rank, fc.tile_world};
  const bool blocks_cached = sc->blocks_valid && sc->blocks_key == blocks_key;
  std::vector<int2>& blocks = sc->h_blocks;
  if (!blocks_cached) {
    blocks.clear();
    sc->blocks_valid = false;
    const bool tiled = fc.tile_world > 1;
    const bool aligned = tiled && (fc.tile_size % BLK == 0);
    for (int by = y0 / BLK; by * BLK < y1; ++by)
      for (int bx = x0 / BLK; bx * BLK < x1; ++bx) {
        if (aligned) {
          int tile = ((by * BLK) / fc.tile_size) * fc.tiles_x + (bx * BLK) / fc.tile_size;
          if (tile % fc.tile_world != fc.tile_rank) continue;
        }
        blocks.push_back(make_int2(bx, by));
      }
  }
  // rendered samples: pixels of the window that this call's tile selection owns
  int64_t live_samples = 0;
  if (fc.tile_world > 1) {
    const int ts = fc.tile_size;
    for (int ty = y0 / ts; ty * ts < y1; ++ty)
      for (int tx = x0 / ts; tx * ts < x1; ++tx) {
        if ((ty * fc.tiles_x + tx) % fc.tile_world != fc.tile_rank) continue;
        int64_t wx = std::min(x1, (tx + 1) * ts) - std::max(x0, tx * ts);
        int64_t wy = std::min(y1, (ty + 1) * ts) - std::max(y0, ty * ts);
        live_samples += wx * wy * spp;
      }
  } else {
    live_samples = (int64_t)(x1 - x0) * (y1 - y0) * spp;
  }
  // With tiles that are whole multiples of the 8x8 blocks the block list already holds exactly the
  // owned pixels, and the kernels need no per-pixel ownership test (two integer divisions per pixel).
  if (fc.tile_world > 1 && fc.tile_size % BLK == 0) {
    fc.tile_world = 1;
    fc.tile_size = 0;
  }
  const size_t samples_per_block = (size_t)BLK_PIX * spp;
  const size_t blocks_per_batch = std::max<size_t>(1, max_batch_samples() / samples_per_block);
  const size_t max_batch_blocks = std::min(blocks_per_batch, std::max<size_t>(blocks.size(), 1));
  int32_t rc = RTMI_OK;

  hipStream_t st = sc->stream;
  HIP_TRY(hipEventRecord(sc->ev_a, st));
  FrameUploads uploads;  // flushed before the first kernel that reads what it carries
  struct UploadGuard {  // an error return before the flush: the device does not hold what the host-side copies say
    rtmi_scene* sc;
    FrameUploads* up;
    ~UploadGuard() {
      if (up->segs.empty()) return;
      sc->genf_uploaded.clear();
      sc->fast.consts_uploaded.clear();
      sc->sargs_uploaded.clear();
    }
  } upload_guard{sc, &uploads};
  uploads.zero(sc->wb.cnt, sizeof(DevCounters));

  // FP32 pre-test of k_gen (scenes without top-level triangle runs: there every sample goes on to
  // the triangle search anyway).  d' = (ul - cam) + uComponent - vComponent in FP32 differs from
  // the FP64 direction by at most rel = sqrt(3)*8*2^-24*B/dmin in relative terms (B bounds the
  // component magnitudes, dmin is the distance from the camera to the image plane), its squared
  // cosine to a unit vector by a few times that; `margin` is far above it.  An object is cleared
  // for a sample only if cos^2 < 1 - (r/|m|)^2 - margin, i.e. the direction misses the sphere
  // around the object by a visible angle; everything closer is decided in FP64 as before.
  fc.n_genf = 0;
  fc.genf = sc->d_genf;
  if (sc->groups[0].n == 0 && sc->dev.n_sphere_like > 0) {
    const double base[3] = {fc.ul[0] - fc.cam[0], fc.ul[1] - fc.cam[1], fc.ul[2] - fc.cam[2]};
    const double dlen = sqrt(cam->dir[0] * cam->dir[0] + cam->dir[1] * cam->dir[1] + cam->dir[2] * cam->dir[2]);
    const double dmin = fabs(cam->focal_length) * dlen;
    const double B = std::max(fabs(base[0]), std::max(fabs(base[1]), fabs(base[2]))) + 20.0;
    const double rel = 1.7320508 * 8.0 * 5.9604644775390625e-08 * B / dmin;
    const double margin = 8.0 * rel + 1.0e-5;
    if (std::isfinite(margin) && margin < 0.05) {
      std::vector<float4> gf;
      for (const rtmi_object& ob : sc->objects) {
        if (ob.kind == RTMI_TRIANGLES) continue;
        const double* c = ob.kind == RTMI_LIGHT ? sc->h_lights[(size_t)ob.first].loc : ob.center;
        const double r = ob.kind == RTMI_LIGHT ? 0.0 : fabs(ob.radius);
        const double m[3] = {c[0] - fc.cam[0], c[1] - fc.cam[1], c[2] - fc.cam[2]};
        const double ml = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        float4 e = make_float4(0.f, 0.f, 0.f, -1.f);  // w <= 0: never cleared
        if (std::isfinite(ml) && ml > r * 1.001 && ml > 0) {
          const double c2 = 1.0 - (r / ml) * (r / ml) - margin;
          if (c2 > 1.0e-3) e = make_float4((float)(m[0] / ml), (float)(m[1] / ml), (float)(m[2] / ml), (float)(c2 * (1.0 - 1.0e-6)));
        }
        gf.push_back(e);
      }
      if (gf.size() != sc->genf_uploaded.size() || memcmp(gf.data(), sc->genf_uploaded.data(), sizeof(float4) * gf.size()) != 0) {
        sc->genf_uploaded = gf;  // stays untouched until the next upload: the async copy reads it later
        uploads.add(sc->d_genf, sc->genf_uploaded.data(), sizeof(float4) * gf.size());
      }
      for (int a = 0; a < 3; ++a) fc.base_f[a] = (float)base[a];
      fc.n_genf = (int)gf.size();
    }
  }
  // The frame without a ray-generation launch (kernels_fast.h): pixels only, every member of the list a bounding sphere
  // without a material, whole-number specular exponents, the reference's hit policy, the cone tree.
  bool fast = false;
  double dir_err = 0.0;
  if (fast_enabled() && use_tree && !nearest && !ext && !sc->no_alloc && sc->inert_secondary && !sc->lean_may_defer &&
      (p->flags & RTMI_FLAG_NO_RAY_STATISTICS) != 0 && singles_enabled() && fc.n_genf > 0 && fc.n_genf == sc->dev.n_objects &&
      sc->dev.n_objects <= FAST_MAX_OBJECTS && sc->n_materials <= FAST_MAX_MATERIALS && sc->n_lights <= FAST_LDS_LIGHTS) {
    dir_err = fast_dir_err(fc, cam);
    fast = dir_err < 1.0e-4;
    for (const TriGroup& g : sc->groups) fast = fast && g.n < (1 << 30);
  }
  FastArgs fa{};
  if (fast) {
    const size_t consts_bytes = sizeof(rtx32::LightF) * std::max(1, sc->n_lights) + sizeof(rtx32::MatF) * std::max(1, sc->n_materials) +
                                sizeof(SphereF) * sc->objects.size();
    rc = ensure_fast_buffers(sc, fast_live_capacity(sc, max_batch_blocks), samples_per_block, max_batch_blocks, consts_bytes);
    if (rc) return rc;
    // the frame's constants for the FP32 lighting and the FP32 sphere test: FP64 differences and products, rounded once
    std::vector<unsigned char> cb(consts_bytes, 0);
    auto* lf = reinterpret_cast<rtx32::LightF*>(cb.data());
    auto* mf = reinterpret_cast<rtx32::MatF*>(cb.data() + sizeof(rtx32::LightF) * std::max(1, sc->n_lights));
    auto* sf = reinterpret_cast<SphereF*>(reinterpret_cast<unsigned char*>(mf) + sizeof(rtx32::MatF) * std::max(1, sc->n_materials));
    auto up = [](double v) { return (float)(fabs(v) * 1.0000002); };  // a float that is not below |v|
    for (int li = 0; li < sc->n_lights; ++li) {
      const rtmi_light& L = sc->h_lights[(size_t)li];
      const double lc[3] = {L.loc[0] - fc.cam[0], L.loc[1] - fc.cam[1], L.loc[2] - fc.cam[2]};
      lf[li].lx = (float)lc[0]; lf[li].ly = (float)lc[1]; lf[li].lz = (float)lc[2];
      lf[li].l1 = up(fabs(lc[0]) + fabs(lc[1]) + fabs(lc[2]));
      lf[li].dr = (float)L.diffuse[0]; lf[li].dg = (float)L.diffuse[1]; lf[li].db = (float)L.diffuse[2];
      lf[li].dmax = up(std::max(fabs(L.diffuse[0]), std::max(fabs(L.diffuse[1]), fabs(L.diffuse[2]))));
      lf[li].sr = (float)L.specular[0]; lf[li].sg = (float)L.specular[1]; lf[li].sb = (float)L.specular[2];
      lf[li].smax = up(std::max(fabs(L.specular[0]), std::max(fabs(L.specular[1]), fabs(L.specular[2]))));
    }
    for (int mi = 0; mi < sc->n_materials; ++mi) {
      const rtmi_material& M = sc->h_materials[(size_t)mi];
      const double ka[3] = {M.Ka[0] * fc.ambient[0], M.Ka[1] * fc.ambient[1], M.Ka[2] * fc.ambient[2]};  // RTF:193-196
      mf[mi].ar = (float)ka[0]; mf[mi].ag = (float)ka[1]; mf[mi].ab = (float)ka[2];
      mf[mi].amax = up(std::max(fabs(ka[0]), std::max(fabs(ka[1]), fabs(ka[2]))));
      mf[mi].dr = (float)M.Kd[0]; mf[mi].dg = (float)M.Kd[1]; mf[mi].db = (float)M.Kd[2];
      mf[mi].dmax = up(std::max(fabs(M.Kd[0]), std::max(fabs(M.Kd[1]), fabs(M.Kd[2]))));
      mf[mi].sr = (float)M.Ks[0]; mf[mi].sg = (float)M.Ks[1]; mf[mi].sb = (float)M.Ks[2];
      mf[mi].smax = up(std::max(fabs(M.Ks[0]), std::max(fabs(M.Ks[1]), fabs(M.Ks[2]))));
      mf[mi].ns = (M.Ns >= 1.0 && M.Ns <= 64.0 && M.Ns == floor(M.Ns)) ? (int)M.Ns : 0;
    }
    for (size_t k = 0; k < sc->objects.size(); ++k) {
      const rtmi_object& ob = sc->objects[k];
      const double cte[3] = {fc.cam[0] - ob.center[0], fc.cam[1] - ob.center[1], fc.cam[2] - ob.center[2]};
      const double kk = (cte[0] * cte[0] + cte[1] * cte[1] + cte[2] * cte[2]) - ob.radius * ob.radius;
      sf[k].cx = (float)cte[0]; sf[k].cy = (float)cte[1]; sf[k].cz = (float)cte[2];
      sf[k].k = (float)kk;
      sf[k].ea = up((dir_err + 8.0 * 5.9604644775390625e-08) * (fabs(cte[0]) + fabs(cte[1]) + fabs(cte[2])));
      sf[k].kabs = up(kk);
      if (!std::isfinite(sf[k].k) || !std::isfinite(sf[k].ea)) sf[k].ea = INFINITY;  // never certain: the FP64 test decides
    }
    if (cb != sc->fast.consts_uploaded) {
      sc->fast.consts_uploaded = cb;  // stays untouched until the next upload: the async copy reads it later
      uploads.add(sc->fast.consts, sc->fast.consts_uploaded.data(), consts_bytes);
    }
    fa.dir_err = (float)(dir_err * 1.0000002);
    fa.n_materials = std::max(1, sc->n_materials);
    fa.hit_id = sc->fast.hit_id;
    fa.rgbf = sc->fast.rgbf;
    fa.lights = reinterpret_cast<const rtx32::LightF*>(sc->fast.consts);
    fa.mats = reinterpret_cast<const rtx32::MatF*>(sc->fast.consts + (reinterpret_cast<unsigned char*>(mf) - cb.data()));
    fa.spheres = reinterpret_cast<const SphereF*>(sc->fast.consts + (reinterpret_cast<unsigned char*>(sf) - cb.data()));
    fa.cap_live = (int)std::min<size_t>(sc->fast.cap_samples / samples_per_block, std::min<size_t>(sc->fast.cap_live, 0x7fffffff / samples_per_block));
    fa.blocks = sc->fast.live_blocks;
    fa.block_slot = sc->fast.block_slot;
    fa.fix_list = sc->fast.fix_list;
    fa.fix_cap = (int)std::min<size_t>(sc->fast.cap_fix, 0x7fffffff);
    fa.force = fast_force();
    fa.ns_uniform = sc->n_materials > 0 ? mf[0].ns : 0;
    for (int mi = 1; mi < sc->n_materials; ++mi)
      if (mf[mi].ns != mf[0].ns) fa.ns_uniform = 0;
  }
  // the general path's per-sample arrays are not needed by a frame of the other path; the block list is
  rc = ensure_buffers(sc, fast ? 0 : max_batch_blocks * samples_per_block, std::max<size_t>(blocks.size(), 1));
  if (rc) return rc;
  sc->dev.n_samples = (int)std::min<size_t>(sc->wb.cap_samples, 0x7fffffff);
  if (!blocks_cached || !sc->blocks_valid) {
    if (!blocks.empty())
      HIP_TRY(hipMemcpyAsync(sc->wb.blocks, blocks.data(), sizeof(int2) * blocks.size(), hipMemcpyHostToDevice, st));
    sc->blocks_key = blocks_key;
    sc->blocks_valid = true;
  }
  // blocks none of whose samples can come near any object (same operands as k_gen's pre-test): every frame,
  // the test is one thread per block
  fc.block_live = nullptr;
  if (fc.n_genf > 0 && !blocks.empty() && !fast) {  // (the other path culls and compacts batch by batch)
    if ((rc = uploads.flush(st))) return rc;
    hipLaunchKernelGGL(k_block_cull, dim3((unsigned)((blocks.size() + 255) / 256)), dim3(256), 0, st, fc,
                       (const int2*)sc->wb.blocks, (int)blocks.size(), sc->wb.block_live);
    fc.block_live = sc->wb.block_live;
  }

  // Per-frame preparation that depends only on the camera and the frame geometry is kept from the
  // previous frame when those are unchanged (a viewer that re-renders with other lights, or a
  // benchmark loop): the sample tables, and each group's search operands for this camera position.
  {
    std::array<double, 18> key = {fc.ul[0], fc.ul[1], fc.ul[2], fc.u[0], fc.u[1], fc.u[2], fc.v[0], fc.v[1], fc.v[2],
                                  fc.ppw_h, fc.ppw_v, fc.sub, (double)fc.W, (double)fc.H, (double)fc.n1,
                                  (double)(uintptr_t)fc.utab, 0.0, 0.0};
    if (!sc->tabs_valid || memcmp(key.data(), sc->tabs_key.data(), sizeof(double) * key.size()) != 0) {
      int ntab = (p->width + p->height) * n1;
      hipLaunchKernelGGL(k_frame_tables, dim3((ntab + 255) / 256), dim3(256), 0, st, fc, sc->wb.tabs,
                         sc->wb.tabs + (size_t)p->width * n1 * 3, const_cast<float*>(fc.utab_f), const_cast<float*>(fc.vtab_f));
      sc->tabs_key = key;
      sc->tabs_valid = true;
    }
  }
  d3 camera = rtx::ld3(cam->pos);
  for (auto& g : sc->groups) {
    const int want = use_tree ? 1 : 2;
    // (margins widened for a larger direction error than this frame's are still conservative)
    if (g.packed_for == want && memcmp(g.packed_cam, cam->pos, sizeof(g.packed_cam)) == 0 && (!use_tree || g.packed_err >= dir_err)) continue;
    if ((rc = use_tree ? pack_group_tree(sc, g, camera, dir_err) : pack_group(sc, g, camera))) return rc;
    g.packed_err = dir_err;
    g.packed_for = want;
    memcpy(g.packed_cam, cam->pos, sizeof(g.packed_cam));
  }

  // A scene whose list holds top-level triangle runs and nothing a primary ray could meet besides: the 8x8 blocks that
  // cannot reach any record of the runs' cone tree hold no hit (k_block_entries); k_gen skips them and k_resolve writes
  // them black, as for the blocks that k_block_cull clears on scenes of spheres.
  if (use_tree && !fast && sc->groups[0].n > 0 && sc->dev.n_sphere_like == 0 && !blocks.empty() && block_entries_enabled() &&
      sc->groups[0].tree.n_root == 1 && sc->groups[0].tree.ltop >= 1 && sc->groups[0].n_records < ((size_t)1 << ENTRY_SHIFT)) {
    const TriGroup& g0 = sc->groups[0];
    hipLaunchKernelGGL(k_block_entries, dim3((unsigned)((blocks.size() * ENTRY_K + 255) / 256)), dim3(256), 0, st, fc, (const int2*)sc->wb.blocks,
                       (int)blocks.size(), (int)blocks.size(), 0.0f, (const f2*)g0.d_rec, g0.tree, (int)g0.n_records, sc->wb.cnt, (int*)nullptr,
                       sc->wb.block_live, 0);
    fc.block_live = sc->wb.block_live;
  }

  // argument blocks of the search launches (one per triangle group), uploaded once per frame
  for (size_t gi = 0; gi < sc->groups.size(); ++gi) {
    const TriGroup& g = sc->groups[gi];
    SearchArgs& a = sc->h_sargs[gi];
    a.queue = gi == 0 ? sc->wb.q_top : sc->wb.q_child;
    a.nq_ptr = gi == 0 ? &sc->wb.cnt->q_top : &sc->wb.cnt->q_child;
    a.npacked_ptr = g.d_npacked;
    a.edge = g.d_edge;
    a.packed_tri = g.d_packed_tri;
    a.dirs = sc->wb.dirs;
    a.tval = sc->wb.tval;
    a.hit_id = fast ? sc->fast.hit_id : sc->wb.hit_id;
    a.tri_hit = sc->wb.tri_hit;
    a.q_shade = sc->wb.q_shade;
    a.cnt = sc->wb.cnt;
    a.tris = sc->d_tris;
    a.planes = sc->d_planes;
    a.cam[0] = fc.cam[0]; a.cam[1] = fc.cam[1]; a.cam[2] = fc.cam[2];
    a.mode = gi == 0 ? 1 : 0;
    a.obj_filter = g.object;
    a.group = g.d_group;
    a.best_pos = fast ? sc->fast.best_pos : sc->wb.best_pos;
    a.cand = fast ? sc->fast.cand : sc->wb.cand;
    a.n_group = g.n;
    a.n_samples = (int)std::min<size_t>(fast ? sc->fast.cap_samples : sc->wb.cap_samples, 0x7fffffff);
    a.n_tris = (int)sc->n_tris;
    a.n_records = (int)g.n_records;
    a.cand_cap = (int)std::min<size_t>(fast ? sc->fast.cap_cand : sc->wb.cap_cand, 0x7fffffff);
    a.multi = fast ? sc->fast.multi : sc->wb.multi;
    a.multi_cap = (int)std::min<size_t>(fast ? sc->fast.cap_multi : sc->wb.cap_multi, 0x7fffffff);
    a.nearest = nearest ? 1 : 0;
    // bounded groups under the reference's hit policy, shaded by k_shade_candidates (which decides the single
    // candidates itself); ~position must stay apart from every position
    a.singles = (use_tree && gi != 0 && !nearest && !ext && singles_enabled() && g.n < (1 << 30)) ? 1 : 0;
    a.tree = g.tree;
  }
  if (sc->sargs_uploaded.size() != sc->h_sargs.size() ||
      memcmp(sc->sargs_uploaded.data(), sc->h_sargs.data(), sizeof(SearchArgs) * sc->h_sargs.size()) != 0) {
    // staged through a copy that stays untouched until the next upload: the async copy reads it later
    sc->sargs_uploaded = sc->h_sargs;
    uploads.add(sc->d_sargs, sc->sargs_uploaded.data(), sizeof(SearchArgs) * sc->groups.size());
  }
  if ((rc = uploads.flush(st))) return rc;

  ShadeConst shc;
  for (int a = 0; a < 3; ++a) { shc.cam[a] = fc.cam[a]; shc.ambient[a] = fc.ambient[a]; }

  long long launches = 0;
  int launch_parity = 0;  // tree search launches of the frame alternate between two sets of list counters
  const bool has_top_runs = sc->groups[0].n > 0;
  sc->ev_used = 0;
  auto next_events = [&](hipEvent_t** ev) -> int32_t {
    if (sc->ev_used == sc->ev_pool.size()) {
      if (sc->no_alloc) {  // cannot happen: one triple per triangle group exists since the scene was created
        g_last_error = "rtmi_render_device_noalloc: out of timing events";
        return RTMI_ERR_UNSUPPORTED;
      }
      std::array<hipEvent_t, 4> t{};
      for (hipEvent_t& e : t) HIP_TRY(hipEventCreate(&e));
      sc->ev_pool.push_back(t);
    }
    *ev = sc->ev_pool[sc->ev_used].data();
    ++sc->ev_used;
    return RTMI_OK;
  };
  bool shade_queue_launched = false;

  for (size_t b0 = 0; b0 < blocks.size(); b0 += blocks_per_batch) {
    size_t nb = std::min(blocks_per_batch, blocks.size() - b0);
    int nsamples = (int)(nb * samples_per_block);
    const int2* d_blk = sc->wb.blocks + b0;
    FrameConst fcb = fc;  // the block indices of a batch's kernels count from the batch's first block
    if (fc.block_live) fcb.block_live = fc.block_live + b0;
    if (b0 > 0) HIP_TRY(hipMemsetAsync(&sc->wb.cnt->q_child, 0, sizeof(int) * 3, st));  // the first batch starts from the cleared counters
    int grid = (nsamples + 255) / 256;
    if (fast) {
      // one wave per group of R x 64 samples that the arrays have room for (the live blocks are at most that many)
      const long long room = std::min<long long>((long long)fa.cap_live * (long long)samples_per_block, nsamples);
      fa.all_blocks = d_blk;
      fa.nsamples = 0;  // the kernels read the live count from the device
      if (b0 > 0) {
        HIP_TRY(hipMemsetAsync(&sc->wb.cnt->fix_n[0][0], 0, sizeof(sc->wb.cnt->fix_n), st));
        HIP_TRY(hipMemsetAsync(&sc->wb.cnt->n_live, 0, sizeof(int), st));
      }
      hipLaunchKernelGGL(k_block_cull_compact, dim3((unsigned)((nb + CULL_BLOCK - 1) / CULL_BLOCK)), dim3(CULL_BLOCK), 0, st, fcb, d_blk, (int)nb,
                         sc->fast.live_blocks, sc->fast.block_slot, fa.cap_live, sc->wb.cnt);
      for (size_t gi = 1; gi < sc->groups.size(); ++gi) {
        TriGroup& g = sc->groups[gi];
        fa.obj = g.object;
        fa.init_hit = gi == 1 ? 1 : 0;
        fa.pos_plane = g.d_pos_plane;
        fa.pos_mt = g.d_pos_mt;
        const int SRf = search_rays_per_lane(true, nsamples, true, (long long)g.n * 64 > (long long)nsamples);
        fa.entries = nullptr;
        fa.entry_masks = 0;
        if (block_entries_enabled() && g.tree.n_root == 1 && g.n_records < ((size_t)1 << ENTRY_SHIFT) && g.tree.ltop >= 1) {
          fa.entries = sc->fast.entries;
          // (a leaf record's entry holds its index and a mask of its triangles: trees of up to 4 M triangles)
          // (not for a mesh so fine that a block reaches most triangles of its records: the masks cost more than they save;
          // and only the walk with four rays per lane reads them)
          fa.entry_masks = leaf_masks_mode() != 0 && SRf == 4 && (size_t)g.tree.n_slots / 8 < ((size_t)1 << ENTRY_LEAF_BITS) &&
                                   (leaf_masks_mode() == 2 || !((long long)g.n * 64 > (long long)nsamples))
                               ? 1 : 0;
          hipLaunchKernelGGL(k_block_entries, dim3((unsigned)(((size_t)fa.cap_live * ENTRY_K + 255) / 256)), dim3(256), 0, st, fcb,
                             (const int2*)sc->fast.live_blocks, -1, fa.cap_live, fa.dir_err, (const f2*)g.d_rec, g.tree, (int)g.n_records, sc->wb.cnt,
                             sc->fast.entries, (unsigned char*)nullptr, fa.entry_masks);
        }
        const int fgrid = (int)std::max<long long>(1, (room + 256 * SRf - 1) / (256 * SRf));
        auto tree_fast = SRf == 8 ? k_search_tree_fast<8> : (SRf == 2 ? k_search_tree_fast<2> : k_search_tree_fast<4>);
        hipEvent_t* ev;
        if ((rc = next_events(&ev))) return rc;
        HIP_TRY(hipEventRecord(ev[0], st));
        const int par = launch_parity;
        launch_parity ^= 1;
        hipLaunchKernelGGL(tree_fast, dim3(fgrid), dim3(256), 0, st, sc->d_sargs + gi, (const f2*)g.d_rec, (const float*)g.d_tedge,
                           (const int*)g.d_slot_pos, par, fcb, fa);
        HIP_TRY(hipEventRecord(ev[3], st));
        hipLaunchKernelGGL(k_search_exact_multi_fast, dim3(sc->exact_grid), dim3(256), 0, st, sc->d_sargs + gi, par, fcb, fa);
        HIP_TRY(hipEventRecord(ev[1], st));
        hipLaunchKernelGGL(k_shade_candidates_fast, dim3(sc->fast_shade_grid), dim3(256), 0, st, fcb, sc->dev, fa, sc->d_sargs + gi, par);
        HIP_TRY(hipEventRecord(ev[2], st));
        ++launches;
      }
      const int npix_f = (int)(nb * BLK_PIX);
      hipLaunchKernelGGL(k_resolve_fast, dim3((npix_f + 255) / 256), dim3(256), 0, st, fcb, fa, npix_f, sc->wb.cnt, (unsigned char*)rgba_dev,
                         (int*)hit_dev);
      hipLaunchKernelGGL(k_fix_pixels, dim3(sc->exact_grid), dim3(256), 0, st, fcb, shc, sc->dev, fa, npix_f, sc->wb.cnt, (unsigned char*)rgba_dev);
      HIP_TRY(hipGetLastError());
      continue;
    }
    hipLaunchKernelGGL(k_gen, dim3((nsamples + 256 * GEN_ITERS - 1) / (256 * GEN_ITERS)), dim3(256), 0, st, fcb, sc->dev, d_blk, nsamples, sc->wb.hit_id, sc->wb.dirs, sc->wb.q_child,
                       sc->wb.q_shade, sc->wb.q_top, sc->wb.cnt, sc->wb.rgb);
    const int SR = search_rays_per_lane(use_tree, nsamples);
    const int search_grid = (nsamples + 256 * SR - 1) / (256 * SR);
    auto scan_kernel = SR == 8 ? k_search_primary<8> : (SR == 2 ? k_search_primary<2> : k_search_primary<4>);
    auto tree_kernel = SR == 8 ? k_search_tree<8> : (SR == 2 ? k_search_tree<2> : k_search_tree<4>);
    // the triangle search of one group for the rays queued for it
    auto search_group = [&](size_t gi) -> int32_t {
      TriGroup& g = sc->groups[gi];
      hipEvent_t* ev;
      int32_t rc2 = next_events(&ev);
      if (rc2) return rc2;
      HIP_TRY(hipEventRecord(ev[0], st));
      if (use_tree) {
        const int par = launch_parity;
        launch_parity ^= 1;
        hipLaunchKernelGGL(tree_kernel, dim3(search_grid), dim3(256), 0, st, sc->d_sargs + gi, (const f2*)g.d_rec,
                           (const float*)g.d_tedge, (const int*)g.d_slot_pos, par);
        HIP_TRY(hipEventRecord(ev[3], st));
        if (nearest) {
          hipLaunchKernelGGL(k_search_exact<true>, dim3(sc->exact_grid), dim3(256), 0, st, sc->d_sargs + gi, par);
          hipLaunchKernelGGL(k_search_nearest, dim3(sc->exact_grid), dim3(256), 0, st, sc->d_sargs + gi, par);
        } else if (sc->h_sargs[gi].singles) {
          hipLaunchKernelGGL(k_search_exact_multi, dim3(sc->exact_grid), dim3(256), 0, st, sc->d_sargs + gi, par);
        } else {
          hipLaunchKernelGGL(k_search_exact<false>, dim3(sc->exact_grid), dim3(256), 0, st, sc->d_sargs + gi, par);
        }
        HIP_TRY(hipEventRecord(ev[1], st));
        if (gi == 0)
          hipLaunchKernelGGL(k_search_finish, dim3(grid), dim3(256), 0, st, sc->d_sargs + gi, par);
        else {
          // pixels only, and a scene whose secondary rays cannot change one: the instantiation without them
          const bool inert = !ext && sc->inert_secondary && (p->flags & RTMI_FLAG_NO_RAY_STATISTICS) != 0;
#define RTMI_LAUNCH_SHADE_CANDIDATES(kernel)                                                                              \
  hipLaunchKernelGGL(kernel, dim3(sc->exact_grid), dim3(256), 0, st, shc, sc->dev, sc->top_args, (const f2*)sc->d_trec, \
                     (const float*)sc->d_gst2, (const int*)sc->groups[0].d_slot_pos, sc->d_sargs + gi, sc->wb.rgb, par)
          if (ext) {  // every intersection goes to the shade queue
            hipLaunchKernelGGL(k_queue_candidates, dim3(sc->exact_grid), dim3(256), 0, st, shc, sc->dev, sc->d_sargs + gi, par);
          } else if (nearest) {
            if (has_top_runs) RTMI_LAUNCH_SHADE_CANDIDATES((k_shade_candidates<true, false, true>));
            else if (inert) RTMI_LAUNCH_SHADE_CANDIDATES((k_shade_candidates<false, true, true>));
            else RTMI_LAUNCH_SHADE_CANDIDATES((k_shade_candidates<false, false, true>));
          } else {
            if (has_top_runs) RTMI_LAUNCH_SHADE_CANDIDATES((k_shade_candidates<true, false>));
            else if (inert) RTMI_LAUNCH_SHADE_CANDIDATES((k_shade_candidates<false, true>));
            else RTMI_LAUNCH_SHADE_CANDIDATES((k_shade_candidates<false, false>));
          }
#undef RTMI_LAUNCH_SHADE_CANDIDATES
        }
      } else {
        hipLaunchKernelGGL(scan_kernel, dim3(search_grid), dim3(256), 0, st, sc->d_sargs + gi, (const f2*)g.d_cone);
        HIP_TRY(hipEventRecord(ev[3], st));
        HIP_TRY(hipEventRecord(ev[1], st));
      }
      HIP_TRY(hipEventRecord(ev[2], st));
      ++launches;
      return RTMI_OK;
    };
    if (sc->groups[0].n > 0) {
      if ((rc = search_group(0))) return rc;
      hipLaunchKernelGGL(k_resolve_top, dim3((grid + RESOLVE_ITERS - 1) / RESOLVE_ITERS), dim3(256), 0, st, sc->dev, sc->wb.q_top, &sc->wb.cnt->q_top, sc->wb.tri_hit,
                         sc->d_tri_owner, sc->wb.hit_id, sc->wb.q_child, sc->wb.q_shade, sc->wb.cnt, (const double*)sc->wb.tval,
                         (const double*)sc->wb.rgb, nearest ? 1 : 0);
    }
    for (size_t gi = 1; gi < sc->groups.size(); ++gi)
      if ((rc = search_group(gi))) return rc;
    // the shade queue stays empty when the cone tree shades every triangle hit itself and the scene
    // lists nothing else that can be shaded (no sphere or light object, no top-level triangles) and
    // nothing with a material that a reflected ray could find (k_shade_candidates queues those samples)
    if (ext || !(use_tree && !has_top_runs && sc->n_plain_shadeable == 0 && !sc->lean_may_defer)) {
      if (b0 == 0) HIP_TRY(hipEventRecord(sc->ev_c, st));  // timed for the first batch only (frames are one batch as a rule)
#define RTMI_LAUNCH_SHADE(kernel)                                                                                            \
  hipLaunchKernelGGL(kernel, dim3(std::min(grid, 4 * sc->exact_grid)), dim3(256), 0, st, shc, sc->dev, sc->top_args,         \
                     (const f2*)sc->d_trec, (const float*)sc->d_gst2, (const int*)sc->groups[0].d_slot_pos, sc->wb.dirs,     \
                     sc->wb.tval, sc->wb.q_shade, &sc->wb.cnt->q_shade, sc->wb.hit_id, sc->wb.rgb, sc->wb.cnt)
#define RTMI_LAUNCH_TRACE_REPLAY(NEAREST, ...)                                                                              \
  do {                                                                                                                      \
    /* tri_hit and q_top are free once the searches have run: they carry the recorded outcomes */                           \
    hipLaunchKernelGGL((k_trace_top<NEAREST, __VA_ARGS__>), dim3(std::min(grid, 4 * sc->exact_grid)), dim3(256), 0, st, shc, \
                       sc->dev, sc->top_args, (const f2*)sc->d_trec, (const float*)sc->d_gst2,                              \
                       (const int*)sc->groups[0].d_slot_pos, sc->wb.dirs, sc->wb.tval, sc->wb.q_shade,                      \
                       &sc->wb.cnt->q_shade, sc->wb.hit_id, sc->wb.rgb, sc->wb.cnt, sc->wb.tri_hit,                         \
                       (unsigned*)sc->wb.q_top, ext_args, (const f2*)sc->d_brec, (const float*)sc->d_bgst2,                 \
                       (const int*)sc->d_bslot_pos, (const int*)sc->d_bmembers, (const int*)sc->d_bowner);                  \
    hipLaunchKernelGGL((k_shade_replay<NEAREST>), dim3(std::min(grid, 4 * sc->exact_grid)), dim3(256), 0, st, shc, sc->dev, \
                       sc->top_args, (const f2*)sc->d_trec, (const float*)sc->d_gst2, (const int*)sc->groups[0].d_slot_pos, \
                       sc->wb.dirs, sc->wb.tval, sc->wb.q_shade, &sc->wb.cnt->q_shade, sc->wb.hit_id, sc->wb.rgb,           \
                       sc->wb.cnt, (const int*)sc->wb.tri_hit, (const unsigned*)sc->wb.q_top);                              \
  } while (0)
#define RTMI_LAUNCH_SHADE_EXT(NEAREST)                                                                                       \
  hipLaunchKernelGGL((k_shade_ext<NEAREST>), dim3(std::min(grid, 4 * sc->exact_grid)), dim3(256), 0, st, shc, sc->dev,      \
                     sc->top_args, (const f2*)sc->d_trec, (const float*)sc->d_gst2, (const int*)sc->groups[0].d_slot_pos,   \
                     ext_args, (const f2*)sc->d_brec, (const float*)sc->d_bgst2, (const int*)sc->d_bslot_pos,               \
                     (const int*)sc->d_bmembers, (const int*)sc->d_bowner, sc->wb.dirs, sc->wb.tval, sc->wb.q_shade,         \
                     &sc->wb.cnt->q_shade, sc->wb.hit_id, sc->wb.rgb, sc->wb.cnt)
      if (ext && trace_then_shade() && sc->dev.n_lights <= TRACE_MAX_LIGHTS) {
        // the recorder without the walks that cannot happen on this scene and frame
        const bool enter = (ext_flags & ENTER_FLAGS) != 0;
        if (nearest) {
          if (has_top_runs && enter) RTMI_LAUNCH_TRACE_REPLAY(true, true, true, true);
          else if (has_top_runs) RTMI_LAUNCH_TRACE_REPLAY(true, true, true, false);
          else if (enter) RTMI_LAUNCH_TRACE_REPLAY(true, true, false, true);
          else RTMI_LAUNCH_TRACE_REPLAY(true, true, false, false);
        } else {
          if (has_top_runs && enter) RTMI_LAUNCH_TRACE_REPLAY(false, true, true, true);
          else if (has_top_runs) RTMI_LAUNCH_TRACE_REPLAY(false, true, true, false);
          else if (enter) RTMI_LAUNCH_TRACE_REPLAY(false, true, false, true);
          else RTMI_LAUNCH_TRACE_REPLAY(false, true, false, false);
        }
      } else if (ext) {
        if (nearest) RTMI_LAUNCH_SHADE_EXT(true);
        else RTMI_LAUNCH_SHADE_EXT(false);
      } else if (has_top_runs && trace_then_shade() && sc->dev.n_lights <= TRACE_MAX_LIGHTS) {
        if (nearest) RTMI_LAUNCH_TRACE_REPLAY(true, false);
        else RTMI_LAUNCH_TRACE_REPLAY(false, false);
      } else if (nearest) {
        if (has_top_runs) RTMI_LAUNCH_SHADE((k_shade<true, true>));
        else RTMI_LAUNCH_SHADE((k_shade<false, true>));
      } else {
        if (has_top_runs) RTMI_LAUNCH_SHADE((k_shade<true, false>));
        else RTMI_LAUNCH_SHADE((k_shade<false, false>));
      }
#undef RTMI_LAUNCH_SHADE_EXT
#undef RTMI_LAUNCH_TRACE_REPLAY
#undef RTMI_LAUNCH_SHADE
      if (b0 == 0) HIP_TRY(hipEventRecord(sc->ev_d, st));
      shade_queue_launched = true;
    }
    int npix = (int)(nb * BLK_PIX);
    hipLaunchKernelGGL(k_resolve, dim3((npix + 255) / 256), dim3(256), 0, st, fcb, d_blk, npix, sc->wb.hit_id, sc->wb.rgb,
                       (const rtmi_object*)sc->d_objects, sc->dev.n_objects, sc->wb.cnt, (unsigned char*)rgba_dev, (int*)hit_dev);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(sc->ev_b, st));
  HIP_TRY(hipMemcpyAsync(sc->h_cnt, sc->wb.cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
  sc->pend_launches = launches;
  sc->pend_live_samples = live_samples;
  sc->pend_shade_timed = shade_queue_launched;
  sc->pend_fast = fast;
  {
    const size_t blk = sc->wb.cap_blocks * (sizeof(int2) + 1) + sc->wb.cap_tabs * (sizeof(double) + sizeof(float)) + sizeof(DevCounters);
    if (fast) {
      const FastBufs& f = sc->fast;
      sc->pend_work_bytes = (int64_t)(f.cap_samples * (2 * sizeof(int) + sizeof(float4)) + (f.cap_cand + f.cap_multi) * sizeof(int2) +
                                      f.cap_fix * sizeof(int) + f.cap_live * (sizeof(int2) + ENTRY_K * sizeof(int)) + f.cap_slots * sizeof(int) + f.cap_consts + blk);
    } else {
      const WorkBufs& w = sc->wb;
      sc->pend_work_bytes = (int64_t)(w.cap_samples * (6 * sizeof(int) + 7 * sizeof(double)) + (w.cap_cand + w.cap_multi) * sizeof(int2) + blk);
    }
  }
  sc->pend_cam = *cam;
  sc->pend_params = *p;
  sc->pend_rgba = rgba_dev;
  sc->pend_hit = hit_dev;
  sc->pending = true;
  return RTMI_OK;
}

static int32_t render_begin_impl(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                                 void* hit_dev) {
  if (!sc) return RTMI_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(sc->mu);
  return render_enqueue(sc, cam, p, rgba_dev, hit_dev);
}

// Wait for the frame enqueued by render_enqueue and report its counters.  The caller holds sc->mu.
static int32_t render_finish(rtmi_scene* sc, rtmi_counters* counters) {
  if (!sc->pending) {
    g_last_error = "no frame in flight on this scene";
    return RTMI_ERR_INVALID_ARGUMENT;
  }
  sc->pending = false;
  HIP_TRY(hipSetDevice(sc->device));
  HIP_TRY(hipStreamSynchronize(sc->stream));
  // More live 8x8 blocks than the per-sample arrays of the path without a ray-generation launch have room for: the
  // frame is rendered again with arrays for what it needed and a quarter on top (the count is exact for a frame of one batch).
  for (int attempt = 0; sc->pend_fast && sc->h_cnt->live_overflow; ++attempt) {
    const size_t had = sc->fast.cap_live;
    sc->fast.live_wanted = std::max<size_t>(had * 2, (size_t)std::max(sc->h_cnt->n_live, 0) * 5 / 4 + 1);
    if (attempt == 8) {
      g_last_error = "the live-block list kept overflowing";
      return RTMI_ERR_HIP;
    }
    const rtmi_camera cam2 = sc->pend_cam;
    const rtmi_params p2 = sc->pend_params;
    int32_t rc = render_enqueue(sc, &cam2, &p2, sc->pend_rgba, sc->pend_hit);
    if (rc != RTMI_OK) return rc;
    sc->pending = false;
    HIP_TRY(hipStreamSynchronize(sc->stream));
  }
  if (sc->h_cnt->overflow && sc->pend_params.search == RTMI_SEARCH_CONE_TREE) {
    // More candidate pairs than the list holds (rays that pass very many triangle cones, e.g. deep
    // stacks of overlapping triangles): the frame is rendered again with the list scan, which
    // needs no such list and gives the same image.
    if (sc->pend_params.hit_policy == RTMI_HIT_NEAREST) {
      // no list scan under this policy: the lists are made four times as long and the frame is rendered again with the
      // tree, until they hold it (the static array of rtmi_render_device_noalloc cannot grow: that call fails, and
      // its output buffers hold an incomplete frame)
      const rtmi_camera cam2 = sc->pend_cam;
      const rtmi_params p2 = sc->pend_params;
      for (int attempt = 0; sc->h_cnt->overflow; ++attempt) {
        if (sc->no_alloc || attempt == 8 || sc->wb.cap_cand > ((size_t)1 << 31)) {
          g_last_error = "the frame has more candidate pairs than the lists hold, and RTMI_HIT_NEAREST has no list-scan fallback; "
                         "the output buffers hold an incomplete frame";
          return RTMI_ERR_UNSUPPORTED;
        }
        const size_t grown = std::max<size_t>(sc->wb.cap_cand * 4, (size_t)NSLOT * 1024);
        (void)hipFree(sc->wb.cand);
        sc->wb.cand = nullptr;
        sc->wb.cap_cand = 0;
        HIP_TRY(hipMalloc(&sc->wb.cand, sizeof(int2) * grown));
        sc->wb.cap_cand = grown;
        int32_t rc = render_enqueue(sc, &cam2, &p2, sc->pend_rgba, sc->pend_hit);
        if (rc != RTMI_OK) return rc;
        sc->pending = false;
        HIP_TRY(hipStreamSynchronize(sc->stream));
      }
    } else {
    rtmi_params p2 = sc->pend_params;
    p2.search = RTMI_SEARCH_LIST_SCAN;
    const rtmi_camera cam2 = sc->pend_cam;
    int32_t rc = render_enqueue(sc, &cam2, &p2, sc->pend_rgba, sc->pend_hit);
    if (rc != RTMI_OK) return rc;
    sc->pending = false;
    HIP_TRY(hipStreamSynchronize(sc->stream));
    sc->pend_params.search = RTMI_SEARCH_CONE_TREE;
    }
  }
  const DevCounters& hc = *sc->h_cnt;
  if (hc.bug) {
    // an index read from device memory was out of range (checked() in the kernels replaced it and
    // went on): the frame cannot be trusted
    g_last_error = "internal index check " + std::to_string(hc.bug) + " failed on the device; the frame is not valid";
    return RTMI_ERR_HIP;
  }
  float ms_total = 0.f, ms_search = 0.f, ms_shade = 0.f, ms_walk = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms_total, sc->ev_a, sc->ev_b));
  for (size_t i = 0; i < sc->ev_used; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_pool[i][0], sc->ev_pool[i][1]));
    ms_search += ms;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_pool[i][0], sc->ev_pool[i][3]));
    ms_walk += ms;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_pool[i][1], sc->ev_pool[i][2]));
    ms_shade += ms;
  }
  if (sc->pend_shade_timed) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_c, sc->ev_d));
    ms_shade += ms;
  }
  if (counters) {
    const int64_t live_samples = sc->pend_live_samples;
    unsigned long long sum[C_NCOUNTERS] = {0};
    for (int sl = 0; sl < NSLOT; ++sl)
      for (int c = 0; c < C_NCOUNTERS; ++c) sum[c] += hc.slot[sl][c];
    const int64_t secondary = (int64_t)(sum[C_REFL] + sum[C_SHADOW]);
    memset(counters, 0, sizeof(*counters));
    counters->samples = live_samples;
    counters->rays_reflection = (int64_t)sum[C_REFL];
    counters->rays_shadow = (int64_t)sum[C_SHADOW];
    // pairs actually evaluated: a ray stops counting at its hit
    counters->tri_tests = (int64_t)(sum[C_TESTS_PRIMARY] + sum[C_TESTS_SECONDARY]);
    counters->sphere_tests = (live_samples + secondary) * sc->dev.n_sphere_like;
    counters->tri_exact_tests = (int64_t)sum[C_EXACT];
    counters->primary_hits = (int64_t)sum[C_HITS];
    counters->shadow_rays_elided = (int64_t)sum[C_SHADOW_ELIDED];
    counters->node_tests = (int64_t)sum[C_NODE_TESTS];
    counters->primary_rays_traced = (int64_t)sum[C_PRIMARY_TRACED];
    counters->hits_decided_fp32 = (int64_t)sum[C_FP32_DECIDED];
    counters->pixels_fixed_fp64 = (int64_t)sum[C_FIX_PIXELS];
    counters->tri_candidates = (int64_t)hc.cand_total;
    counters->ms_total = ms_total;
    counters->ms_tri_search = ms_search;
    counters->tri_search_launches = sc->pend_launches;
    counters->tri_tests_primary_kernel = (int64_t)sum[C_TESTS_PRIMARY];
    counters->ms_primary_kernel = ms_walk;
    counters->tri_edge_tests = (int64_t)sum[C_EDGE_TESTS];
    counters->work_buffer_bytes = sc->pend_work_bytes;
    counters->ms_shade = ms_shade;
  }
  return RTMI_OK;
}

static int32_t render_end_impl(rtmi_scene* sc, rtmi_counters* counters) {
  if (!sc) return RTMI_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(sc->mu);
  return render_finish(sc, counters);
}

static int32_t render_impl(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                           void* hit_dev, rtmi_counters* counters) {
  int32_t rc = render_begin_impl(sc, cam, p, rgba_dev, hit_dev);
  if (rc != RTMI_OK) return rc;
  return render_end_impl(sc, counters);
}

// Everything the camera- and geometry-keyed caches say about the work buffers' contents is forgotten
// (the buffers are about to change hands).
static void forget_buffer_caches(rtmi_scene* sc) {
  sc->tabs_valid = false;
  sc->blocks_valid = false;
  sc->sargs_uploaded.clear();
}

// rtmi_render_device with the device's static array in place of the scene's own work buffers.
static int32_t render_noalloc_impl(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                                   void* hit_dev, rtmi_counters* counters) {
  if (!sc) return RTMI_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(sc->mu);
  if (sc->pending) {
    g_last_error = "a frame is already in flight on this scene; call rtmi_render_end first";
    return RTMI_ERR_INVALID_ARGUMENT;
  }
  unsigned char* base = nullptr;
  if (sc->device >= 0 && sc->device < MAX_DEVICES) {
    std::lock_guard<std::mutex> slock(g_scratch_mu);
    base = g_scratch_ptr[sc->device];
  }
  if (!base) {
    g_last_error = "rtmi_render_device_noalloc: no static work array on device " + std::to_string(sc->device);
    return RTMI_ERR_UNSUPPORTED;
  }
  std::lock_guard<std::mutex> turn(g_scratch_busy[sc->device]);
  WorkBufs own = sc->wb;
  sc->wb = carve_scratch(base);
  sc->no_alloc = true;
  forget_buffer_caches(sc);
  int32_t rc = render_enqueue(sc, cam, p, rgba_dev, hit_dev);
  if (rc == RTMI_OK) rc = render_finish(sc, counters);  // a candidate-list overflow renders again, within the same limits
  if (rc != RTMI_OK) {  // whatever was enqueued before the failure is drained before the buffers change back
    (void)hipStreamSynchronize(sc->stream);
    sc->pending = false;
  }
  sc->no_alloc = false;
  sc->wb = own;
  forget_buffer_caches(sc);
  return rc;
}

extern "C" int32_t rtmi_render_begin(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                                     void* hit_dev) {
  return guarded([&] { return render_begin_impl(sc, cam, p, rgba_dev, hit_dev); });
}

extern "C" int32_t rtmi_render_end(rtmi_scene* sc, rtmi_counters* counters) {
  return guarded([&] { return render_end_impl(sc, counters); });
}

extern "C" int32_t rtmi_render_device(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                                      void* hit_dev, rtmi_counters* counters) {
  return guarded([&] { return render_impl(sc, cam, p, rgba_dev, hit_dev, counters); });
}

extern "C" int32_t rtmi_render_device_noalloc(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p,
                                              void* rgba_dev, void* hit_dev, rtmi_counters* counters) {
  return guarded([&] { return render_noalloc_impl(sc, cam, p, rgba_dev, hit_dev, counters); });
}

extern "C" int64_t rtmi_noalloc_max_samples(void) { return (int64_t)NOALLOC_MAX_SAMPLES; }

extern "C" int32_t rtmi_direction_error_bound(const rtmi_camera* cam, int32_t width, int32_t height, double* bound) {
  if (!cam || !bound || width < 2 || height < 2 || width > MAX_FRAME_EDGE || height > MAX_FRAME_EDGE) return RTMI_ERR_INVALID_ARGUMENT;
  FrameConst fc{};
  frame_basis(cam, width, height, &fc);
  *bound = fast_dir_err(fc, cam);
  return RTMI_OK;
}

// Work enqueued on the scene's stream after this call starts only when `hip_event` has completed.
extern "C" int32_t rtmi_scene_wait_event(rtmi_scene* sc, void* hip_event) {
  return guarded([&]() -> int32_t {
    if (!sc || !hip_event) return RTMI_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(sc->mu);
    HIP_TRY(hipSetDevice(sc->device));
    HIP_TRY(hipStreamWaitEvent(sc->stream, static_cast<hipEvent_t>(hip_event), 0));
    return RTMI_OK;
  });
}

static int32_t render_host_impl(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, uint8_t* rgba_out,
                                int32_t* hit_id_out, rtmi_counters* counters) {
  if (!sc || !p || !rgba_out) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width < 2 || p->height < 2) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width > MAX_FRAME_EDGE || p->height > MAX_FRAME_EDGE) return RTMI_ERR_UNSUPPORTED;
  int spp = rtmi_samples_per_pixel(p->pixel_subdivision);
  if (spp < 0) return spp;
  HIP_TRY(hipSetDevice(sc->device));
  size_t nbytes = (size_t)4 * p->width * p->height;
  size_t nhit = (size_t)p->width * p->height * spp;
  unsigned char* d_rgba = nullptr;
  int* d_hit = nullptr;
  HIP_TRY(hipMalloc(&d_rgba, nbytes));
  hipError_t e = hipMemcpy(d_rgba, rgba_out, nbytes, hipMemcpyHostToDevice);  // unrendered pixels keep the caller's bytes
  if (e == hipSuccess && hit_id_out) {
    e = hipMalloc(&d_hit, sizeof(int) * nhit);
    if (e == hipSuccess) e = hipMemcpy(d_hit, hit_id_out, sizeof(int) * nhit, hipMemcpyHostToDevice);
  }
  int32_t rc = RTMI_OK;
  if (e != hipSuccess) {
    g_last_error = std::string("rtmi_render staging: ") + hipGetErrorString(e);
    rc = e == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP;
  }
  if (rc == RTMI_OK) rc = guarded([&] { return render_impl(sc, cam, p, d_rgba, d_hit, counters); });
  if (rc == RTMI_OK) {
    e = hipMemcpy(rgba_out, d_rgba, nbytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess && hit_id_out) e = hipMemcpy(hit_id_out, d_hit, sizeof(int) * nhit, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      g_last_error = std::string("rtmi_render readback: ") + hipGetErrorString(e);
      rc = RTMI_ERR_HIP;
    }
  }
  (void)hipFree(d_rgba);
  (void)hipFree(d_hit);
  return rc;
}

extern "C" int32_t rtmi_render(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, uint8_t* rgba_out,
                               int32_t* hit_id_out, rtmi_counters* counters) {
  return guarded([&] { return render_host_impl(sc, cam, p, rgba_out, hit_id_out, counters); });
}
