// rtmi_tree.h -- cone tree over a triangle group: the default search of primary rays.
// Included by rtmi.hip (after its helpers); not a stand-alone translation unit.
//
// The reference takes the FIRST hit in list order (RayTracingFunctions.swift:41-43, :83,
// :86-93).  "First in list order" is the hit with the lowest list position, so the hits of a
// ray may be found in any order as long as all of them are found and the lowest position is
// kept.  That frees the search from the list order:
//
//   * once per scene the triangles of a group are permuted into spatially compact leaves of
//     eight (median splits of the centroids); eight consecutive leaves form a level-1 node's
//     sibling set, eight of those a level-2 node's, and so on: an implicit, left-complete
//     8-ary tree with no pointers;
//   * once per frame (k_pack_tree) every triangle that a ray from the camera can hit at all
//     gets the same cone and edge operands as in the list scan, and every inner node gets a
//     cone (axis, half-angle) that holds every triangle below it (up to 512 triangles: fitted to
//     their corners; above that: around the cones of its children); culled or absent
//     children have the zero vector (never pass), children wider than about 75 degrees are
//     flagged "always descend";
//   * k_search_tree walks the tree once per wave (all R*64 rays of a wave start at the camera
//     and lie within a few pixels of each other): one record of eight sibling cones is
//     fetched through scalar loads and evaluated with packed FP32 FMAs exactly like a batch of
//     the list scan; a child is entered if ANY ray of the wave passes its cone.  At a leaf the
//     nominated (ray, triangle) pairs go through the edge test; the survivors are the
//     candidates that k_search_exact decides in FP64 (rtx::tri_hit), keeping the lowest list
//     position per ray.
//
// A ray that truly hits a triangle has its direction inside that triangle's cone and inside
// the cone of every ancestor, so it reaches the leaf: the filter never loses a hit.
#pragma once

constexpr int TREE_MAX_LEVELS = 8;  // record levels 0..7: one byte of the traversal word per level
constexpr int TREE_BLOCK = 512;     // slots per k_pack_tree workgroup: 64 leaves, 8 level-2 nodes, 1 level-3 node
constexpr int REC_DWORDS = 32;      // per record: 4 pairs x 3 components x 2 children, "always" mask, padding
constexpr int REC_ALWAYS = 24;      // dword index of the record's "always descend" child mask

struct NodeCone {
  double ax, ay, az;  // unit axis
  double c, s;        // cosine and sine of the half-angle
  int kind;           // 0 empty (nothing below can be hit), 1 cone, 2 wide (always descend)
  int pad;
};

struct TreeDev {
  const int* slot_tri;  // slot -> triangle index, -1 for padding slots
  const int* slot_pos;  // slot -> position in the group's list order
  float* rec;           // records of all levels, level 0 first
  float* edge;          // per slot: EDGE_FLOATS
  NodeCone* nodes;      // cones of the nodes of levels >= 3 (inputs of the upper merges)
  unsigned int* ticket; // workgroups of k_pack_tree that have finished their part
  int n_slots;          // multiple of 8
  int ltop;             // level of the root record(s)
  int n_root;           // number of root records (1 unless the group exceeds 8^8 slots)
  int rec_off[TREE_MAX_LEVELS];        // first record of each level
  int n_nodes[TREE_MAX_LEVELS + 2];    // nodes per level; level 0 = slots
  int node_off[TREE_MAX_LEVELS + 2];   // first NodeCone of levels >= 3
};

// ---------------------------------------------------------------------------------------
// host: spatial order and level table
// ---------------------------------------------------------------------------------------
struct TreeSplitter {
  const std::vector<std::array<double, 3>>& cen;
  std::vector<int>& items;     // positions being permuted
  std::vector<int>& slot_pos;  // result
  void run(size_t lo, size_t hi, size_t leaf_lo, size_t leaf_cnt) {
    if (leaf_cnt <= 1) {
      for (size_t i = lo; i < hi; ++i) slot_pos[leaf_lo * 8 + (i - lo)] = items[i];
      return;
    }
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = lo; i < hi; ++i)
      for (int a = 0; a < 3; ++a) {
        double c = cen[items[i]][a];
        mn[a] = std::min(mn[a], c);
        mx[a] = std::max(mx[a], c);
      }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
      if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    // left part: the largest power of two of whole leaves that leaves something for the right,
    // so that every aligned run of 8^k leaves is one subtree
    size_t left_leaves = 1;
    while (left_leaves * 2 < leaf_cnt) left_leaves *= 2;
    size_t left_count = left_leaves * 8;
    std::nth_element(items.begin() + lo, items.begin() + lo + left_count, items.begin() + hi, [&](int a, int b) {
      double ca = cen[a][axis], cb = cen[b][axis];
      return ca < cb || (ca == cb && a < b);
    });
    run(lo, lo + left_count, leaf_lo, left_leaves);
    run(lo + left_count, hi, leaf_lo + left_leaves, leaf_cnt - left_leaves);
  }
};

struct TreeHost {
  std::vector<int> slot_pos, slot_tri;
  TreeDev dev{};
  size_t n_records = 0, n_node_buf = 0;
};

// idx: list position -> triangle index
static void tree_build_host(const rtmi_triangle* tris, const std::vector<int>& idx, TreeHost* th) {
  const size_t n = idx.size();
  const size_t n_leaf = std::max<size_t>(1, (n + 7) / 8);
  th->slot_pos.assign(n_leaf * 8, -1);
  if (n) {
    std::vector<std::array<double, 3>> cen(n);
    for (size_t p = 0; p < n; ++p) {
      const rtmi_triangle& t = tris[idx[p]];
      for (int a = 0; a < 3; ++a) {
        double c = (t.v[0][a] + t.v[1][a] + t.v[2][a]) / 3.0;
        cen[p][a] = std::isfinite(c) ? c : 0.0;
      }
    }
    std::vector<int> items(n);
    for (size_t p = 0; p < n; ++p) items[p] = (int)p;
    TreeSplitter sp{cen, items, th->slot_pos};
    sp.run(0, n, 0, n_leaf);
  }
  th->slot_tri.assign(n_leaf * 8, -1);
  for (size_t s = 0; s < th->slot_pos.size(); ++s)
    if (th->slot_pos[s] >= 0) th->slot_tri[s] = idx[th->slot_pos[s]];

  TreeDev& d = th->dev;
  d.n_slots = (int)(n_leaf * 8);
  for (int l = 0; l < TREE_MAX_LEVELS + 2; ++l) d.n_nodes[l] = d.node_off[l] = 0;
  d.n_nodes[0] = d.n_slots;
  size_t rec = 0, nodes = 0;
  d.ltop = -1;
  for (int l = 0; l < TREE_MAX_LEVELS; ++l) {
    int n_rec = (d.n_nodes[l] + 7) / 8;
    d.rec_off[l] = (int)rec;
    if (d.ltop < 0) {
      rec += (size_t)n_rec;
      d.n_nodes[l + 1] = n_rec;  // one parent per record
      if (n_rec == 1 || l == TREE_MAX_LEVELS - 1) {
        d.ltop = l;
        d.n_root = n_rec;
      }
    }
  }
  for (int l = 3; l <= d.ltop + 1; ++l) {
    d.node_off[l] = (int)nodes;
    nodes += (size_t)d.n_nodes[l];
  }
  th->n_records = rec;
  th->n_node_buf = std::max<size_t>(nodes, 1);
}

// Bounding-sphere records over the same slot order, for rays of ANY origin (reflection and shadow
// rays against the top-level runs).  Built once per scene on the host in FP64.  A record holds eight
// sibling spheres as 4 pairs x (cx, cy, cz, nK) x (child a, child b): a ray's LINE comes within rho
// of c iff (d.(c-o))^2 + rho^2 - |c-o|^2 >= 0, evaluated as T + nK >= 0 with A = d.c - d.o,
// T = A*A + (2 c.o - |o|^2) and nK = rho^2 + delta - |c|^2; delta = 64*2^-24*(|c|_1 + r_o1)^2 exceeds the
// FP32 evaluation error of T (about 40*2^-24*(|c|_1+|o|_1)^2).  Coordinates are relative to c0; r_o1
// bounds |o - c0|_1 over every possible ray origin.  Absent children get nK = -3e38 (never pass),
// non-finite ones +3e38 (always pass).  A node's sphere sits at the mean of its children's centres and reaches the
// farthest corner of the triangles below it.
//
// A second block of th.n_records records follows the first, in the same layout: per sphere a cone of normals
// (ax, ay, az, thr) over the triangles below that node.  A triangle is only hit from its front, normal.direction < 0
// (REN:145-146); if every normal below a node lies within theta < 90 deg of the unit axis a, then
// a.direction >= sin(theta) makes every normal.direction >= 0 and the node can be passed over.  thr = sin(theta) +
// 1e-4 (the margin covers the FP32 evaluation and directions that are unit vectors only to rounding); +3e38 where
// no such cone exists (never passed over).  Filled for levels >= 1; *cone_levels gets a bit per level that has
// at least one usable cone.
static void tree_sphere_records(const rtmi_triangle* tris, const TreeHost& th, const double c0[3], double r_o1,
                                std::vector<float>* out, unsigned* cone_levels) {
  struct Sph { double c[3]; double r; bool valid; };
  const TreeDev& d = th.dev;
  out->assign(2 * th.n_records * REC_DWORDS, 0.0f);
  {
    float* cone = out->data() + th.n_records * REC_DWORDS;
    for (size_t r = 0; r < th.n_records; ++r)
      for (int pr = 0; pr < 4; ++pr) cone[r * REC_DWORDS + pr * 8 + 6] = cone[r * REC_DWORDS + pr * 8 + 7] = 3.0e38f;
    *cone_levels = 0;
    // unit normals per slot (Renderables.swift:136: normalize(cross(v1 - v0, v2 - v0))); absent or degenerate: skipped
    const int ns = d.n_nodes[0];
    std::vector<double> nrm((size_t)ns * 3, 0.0);
    std::vector<char> has((size_t)ns, 0);
    for (int p = 0; p < ns; ++p) {
      const int ti = th.slot_tri[p];
      if (ti < 0) continue;
      double e1[3], e2[3];
      for (int a = 0; a < 3; ++a) { e1[a] = tris[ti].v[1][a] - tris[ti].v[0][a]; e2[a] = tris[ti].v[2][a] - tris[ti].v[0][a]; }
      const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
      const double len = sqrt(cx * cx + cy * cy + cz * cz);
      // a triangle without a finite unit normal is never hit (its normal.direction is NaN): it constrains nothing
      if (!(len > 0) || !std::isfinite(len)) continue;
      nrm[(size_t)p * 3 + 0] = cx / len; nrm[(size_t)p * 3 + 1] = cy / len; nrm[(size_t)p * 3 + 2] = cz / len;
      has[p] = 1;
    }
    long long span = 1;
    for (int L = 1; L <= d.ltop; ++L) {
      span *= 8;
      for (int j = 0; j < d.n_nodes[L]; ++j) {
        const long long p0 = (long long)j * span, p1 = std::min<long long>(p0 + span, ns);
        double ax = 0, ay = 0, az = 0;
        for (long long p = p0; p < p1; ++p) if (has[p]) { ax += nrm[p * 3]; ay += nrm[p * 3 + 1]; az += nrm[p * 3 + 2]; }
        const double al = sqrt(ax * ax + ay * ay + az * az);
        if (!(al > 1e-9)) continue;
        ax /= al; ay /= al; az /= al;
        // least cosine between an axis and the normals below the node, and the normal that has it
        auto least = [&](double x, double y, double z, long long* at) {
          double c = INFINITY;
          *at = -1;
          for (long long p = p0; p < p1; ++p)
            if (has[p]) {
              const double dq = x * nrm[p * 3] + y * nrm[p * 3 + 1] + z * nrm[p * 3 + 2];
              if (dq < c) { c = dq; *at = p; }
            }
          return std::min(c, 1.0);
        };
        long long far = -1;
        double cmin = least(ax, ay, az, &far);
        // the mean normal is not the axis of the narrowest cone: a few steps towards the normal farthest from the axis
        // (nodes of up to 512 triangles), the best axis kept
        double bx = ax, by = ay, bz = az;
        for (int it = 1; it <= (span <= 512 ? 24 : 0) && far >= 0; ++it) {
          bx += (nrm[far * 3] - bx) / (double)(it + 1);
          by += (nrm[far * 3 + 1] - by) / (double)(it + 1);
          bz += (nrm[far * 3 + 2] - bz) / (double)(it + 1);
          const double bl = sqrt(bx * bx + by * by + bz * bz);
          if (!(bl > 1e-9)) break;
          bx /= bl; by /= bl; bz /= bl;
          const double c = least(bx, by, bz, &far);
          if (c > cmin) { cmin = c; ax = bx; ay = by; az = bz; }
        }
        if (!(cmin > 0.1)) continue;  // wider than about 84 degrees: not worth a test
        const double sin_theta = sqrt(std::max(0.0, 1.0 - std::min(cmin, 1.0) * std::min(cmin, 1.0)));
        float* dst = cone + (size_t)(d.rec_off[L] + (j >> 3)) * REC_DWORDS + ((j & 7) >> 1) * 8 + (j & 1);
        dst[0] = (float)ax; dst[2] = (float)ay; dst[4] = (float)az; dst[6] = (float)(sin_theta + 1e-4);
        *cone_levels |= 1u << L;
      }
    }
  }
  // nodes that do not exist (the tail of a level's last record) must never pass: an all-zero entry
  // WOULD pass for a ray whose line runs through c0
  for (size_t r = 0; r < th.n_records; ++r)
    for (int pr = 0; pr < 4; ++pr) {
      (*out)[r * REC_DWORDS + pr * 8 + 6] = -3.0e38f;
      (*out)[r * REC_DWORDS + pr * 8 + 7] = -3.0e38f;
    }
  auto store = [&](int level, int j, const Sph& sp) {
    float* dst = out->data() + (size_t)(d.rec_off[level] + (j >> 3)) * REC_DWORDS + ((j & 7) >> 1) * 8 + (j & 1);
    float fx = 0.f, fy = 0.f, fz = 0.f, fk = -3.0e38f;
    if (sp.valid) {
      const double eps32 = 5.9604644775390625e-08;  // 2^-24
      double s1 = fabs(sp.c[0]) + fabs(sp.c[1]) + fabs(sp.c[2]) + r_o1;
      double nK = sp.r * sp.r * (1.0 + 1e-6) + 64.0 * eps32 * s1 * s1 - (sp.c[0] * sp.c[0] + sp.c[1] * sp.c[1] + sp.c[2] * sp.c[2]);
      fx = (float)sp.c[0]; fy = (float)sp.c[1]; fz = (float)sp.c[2]; fk = (float)nK;
      if (!std::isfinite(fx) || !std::isfinite(fy) || !std::isfinite(fz) || !std::isfinite(fk)) { fx = fy = fz = 0.f; fk = 3.0e38f; }
    }
    dst[0] = fx; dst[2] = fy; dst[4] = fz; dst[6] = fk;
  };
  std::vector<Sph> cur((size_t)d.n_nodes[0]);
  for (int p = 0; p < d.n_nodes[0]; ++p) {
    Sph sp{};
    const int ti = th.slot_tri[p];
    sp.valid = ti >= 0;
    if (sp.valid) {
      double v[3][3];
      for (int q = 0; q < 3; ++q) for (int a = 0; a < 3; ++a) v[q][a] = tris[ti].v[q][a] - c0[a];
      for (int a = 0; a < 3; ++a) sp.c[a] = (v[0][a] + v[1][a] + v[2][a]) / 3.0;
      double r2 = 0;
      for (int q = 0; q < 3; ++q) {
        double e = 0;
        for (int a = 0; a < 3; ++a) e += (v[q][a] - sp.c[a]) * (v[q][a] - sp.c[a]);
        r2 = std::max(r2, e);
      }
      sp.r = sqrt(r2) * (1.0 + 1e-12);
      if (!(sp.r >= 0)) sp.r = NAN;  // NaN vertices: stored as "always pass"
    }
    store(0, p, sp);
    cur[p] = sp;
  }
  long long span = 1;
  for (int L = 1; L <= d.ltop; ++L) {
    span *= 8;
    std::vector<Sph> next((size_t)d.n_nodes[L]);
    for (int j = 0; j < d.n_nodes[L]; ++j) {
      Sph sp{};
      int nv = 0;
      for (int k = 0; k < 8; ++k) {
        int ch = j * 8 + k;
        if (ch >= d.n_nodes[L - 1] || !cur[ch].valid) continue;
        for (int a = 0; a < 3; ++a) sp.c[a] += cur[ch].c[a];
        ++nv;
      }
      sp.valid = nv > 0;
      if (sp.valid) {
        for (int a = 0; a < 3; ++a) sp.c[a] /= nv;
        bool bad = false;
        for (int k = 0; k < 8; ++k) {
          int ch = j * 8 + k;
          if (ch >= d.n_nodes[L - 1] || !cur[ch].valid) continue;
          if (cur[ch].r != cur[ch].r) bad = true;
        }
        // The sphere holds every CORNER below the node (a sphere that holds the corners holds the triangles): tighter
        // than reaching around the children's spheres, whose slack would compound from level to level.  Its centre
        // starts at the mean of the children's centres and, for nodes of up to 512 triangles, takes a few steps towards
        // the farthest corner (Badoiu and Clarkson's approximation of the smallest enclosing ball); the best centre
        // seen is kept.
        const long long p0 = (long long)j * span, p1 = std::min<long long>(p0 + span, d.n_nodes[0]);
        auto far2 = [&](const double c[3], double far[3]) {  // squared distance to the farthest corner, and that corner
          double r2m = -1.0;
          for (long long p = p0; p < p1; ++p) {
            const int ti = th.slot_tri[p];
            if (ti < 0) continue;
            for (int q = 0; q < 3; ++q) {
              double e = 0, w[3];
              for (int a = 0; a < 3; ++a) {
                w[a] = tris[ti].v[q][a] - c0[a];
                e += (w[a] - c[a]) * (w[a] - c[a]);
              }
              if (e != e) return (double)NAN;
              if (e > r2m) { r2m = e; far[0] = w[0]; far[1] = w[1]; far[2] = w[2]; }
            }
          }
          return r2m < 0 ? 0.0 : r2m;
        };
        double r2 = 0;
        if (!bad) {
          double c[3] = {sp.c[0], sp.c[1], sp.c[2]}, far[3] = {0, 0, 0};
          r2 = far2(c, far);
          if (r2 != r2) bad = true;
          const int steps = span <= 512 ? 24 : 0;
          for (int it = 1; it <= steps && !bad; ++it) {
            for (int a = 0; a < 3; ++a) c[a] += (far[a] - c[a]) / (double)(it + 1);
            const double e = far2(c, far);
            if (e != e || !std::isfinite(c[0] + c[1] + c[2])) break;  // keep what we have
            if (e < r2) { r2 = e; sp.c[0] = c[0]; sp.c[1] = c[1]; sp.c[2] = c[2]; }
          }
        }
        sp.r = bad ? NAN : sqrt(r2) * (1.0 + 1e-12);  // NaN is stored as "always pass"
      }
      store(L, j, sp);
      next[j] = sp;
    }
    cur.swap(next);
  }
}

// ---------------------------------------------------------------------------------------
// device: per-frame operands
// ---------------------------------------------------------------------------------------
// The cone of a node from the cones of its (up to eight) children; *always gets the mask of
// wide children.  A cone around p with cos(half-angle) <= cos(angle(p, a_k) + theta_k) for every
// child k contains every child cone; the cross product gives sin(angle) without cancellation.
__device__ inline NodeCone merge_cones(const NodeCone* ch, int n_valid, unsigned* always) {
  NodeCone r;
  r.ax = r.ay = r.az = 0.0;
  r.c = 1.0;
  r.s = 0.0;
  r.kind = 0;
  r.pad = 0;
  unsigned wide = 0;
  bool any = false;
  d3 sum = rtx::mk(0.0, 0.0, 0.0);
  for (int k = 0; k < 8; ++k) {
    int kind = k < n_valid ? ch[k].kind : 0;
    if (kind == 2) wide |= 1u << k;
    if (kind == 1) {
      any = true;
      sum = rtx::add(sum, rtx::mk(ch[k].ax, ch[k].ay, ch[k].az));
    }
  }
  *always = wide;
  if (wide) { r.kind = 2; return r; }
  if (!any) return r;
  double len = sqrt(rtx::dp(sum, sum));
  if (!(len > 1.0e-3)) { r.kind = 2; return r; }  // axes cancel (or NaN): no useful cone
  d3 p = rtx::divs(sum, len);
  double cmin = 1.0;
  for (int k = 0; k < 8; ++k) {
    if (k >= n_valid || ch[k].kind != 1) continue;
    d3 a = rtx::mk(ch[k].ax, ch[k].ay, ch[k].az);
    double ca = rtx::dp(p, a);
    d3 cr = rtx::cross(p, a);
    double sa = sqrt(rtx::dp(cr, cr));
    cmin = fmin(cmin, ca * ch[k].c - sa * ch[k].s);  // cos(alpha + theta)
  }
  cmin -= 1.0e-12;  // FP64 rounding of the lines above
  if (!(cmin >= 0.25)) { r.kind = 2; return r; }
  r.kind = 1;
  r.ax = p.x; r.ay = p.y; r.az = p.z;
  r.c = cmin;
  r.s = sqrt(fmax(0.0, 1.0 - cmin * cmin));
  return r;
}

// The cone of a node of levels 1 to 3 is fitted to the CORNERS below it instead: a ray that hits a triangle has its
// direction inside the spherical triangle of the three unit vectors from the origin to the corners, and a cone of less
// than 90 degrees that holds the corners holds that triangle.  So with p the node's axis, cos(half-angle) = the least
// p.corner over every kept triangle below the node bounds all of them, without the slack of a cone around cones (which
// reaches to the far side of every child's cone and compounds per level).  node_axis is the first half of merge_cones
// (kinds, "always" mask, axis = normalised sum of the children's axes); node_finish turns the least p.corner into the cone.
__device__ inline NodeCone node_axis(const NodeCone* ch, int n_valid, unsigned* always) {
  NodeCone r;
  r.ax = r.ay = r.az = 0.0;
  r.c = 1.0;
  r.s = 0.0;
  r.kind = 0;
  r.pad = 0;
  unsigned wide = 0;
  bool any = false;
  d3 sum = rtx::mk(0.0, 0.0, 0.0);
  for (int k = 0; k < 8; ++k) {
    int kind = k < n_valid ? ch[k].kind : 0;
    if (kind == 2) wide |= 1u << k;
    if (kind == 1) {
      any = true;
      sum = rtx::add(sum, rtx::mk(ch[k].ax, ch[k].ay, ch[k].az));
    }
  }
  *always = wide;
  if (wide) { r.kind = 2; return r; }
  if (!any) return r;
  double len = sqrt(rtx::dp(sum, sum));
  if (!(len > 1.0e-3)) { r.kind = 2; return r; }  // axes cancel (or NaN): no useful cone
  d3 p = rtx::divs(sum, len);
  r.kind = 1;
  r.ax = p.x; r.ay = p.y; r.az = p.z;
  return r;
}
__device__ inline void node_finish(NodeCone* r, double cmin) {
  if (r->kind != 1) return;
  cmin -= 1.0e-12;  // FP64 rounding of the dot products
  if (!(cmin >= 0.25)) {  // (also NaN)
    r->kind = 2;
    r->ax = r->ay = r->az = 0.0;
    return;
  }
  r->c = cmin;
  r->s = sqrt(fmax(0.0, 1.0 - cmin * cmin));
}
// The least p.corner of one triangle for the axis of node `n` (1 when the triangle is not kept or the node has no cone).
__device__ inline double corner_min(const NodeCone& n, bool has, d3 a, d3 b, d3 c) {
  if (!has || n.kind != 1) return 1.0;
  const d3 p = rtx::mk(n.ax, n.ay, n.az);
  return fmin(rtx::dp(p, a), fmin(rtx::dp(p, b), rtx::dp(p, c)));
}

// Address of child j's x component inside a record; y and z follow at +2 and +4.
__device__ inline float* rec_entry(float* rec, int rec_index, int j) {
  return rec + (size_t)rec_index * REC_DWORDS + (j >> 1) * 6 + (j & 1);
}

// Store the FP32 test vector of a node cone: w = axis / (cos - 2^-19), pass iff d.w >= 1 (the
// margin covers the FP32 evaluation, see k_pack_write); NaN for a wide node, zero for an empty one.
// dir_err: how far the unit directions the search tests may be from the reference's (zero when they are the FP64
// directions rounded to float; see fast_dir_err for the search that derives them in FP32): a direction d with
// d*.axis >= cos still has d.axis >= cos - dir_err.
__device__ inline void store_cone(float* dst, NodeCone* nc, double dir_err) {
  float wx = 0.f, wy = 0.f, wz = 0.f;
  if (nc->kind == 1) {
    double den = nc->c - 1.9073486328125e-06 - dir_err;  // 2^-19
    wx = (float)(nc->ax / den); wy = (float)(nc->ay / den); wz = (float)(nc->az / den);
    if (!(den >= 0.0625) || !isfinite(wx) || !isfinite(wy) || !isfinite(wz)) nc->kind = 2;
  }
  if (nc->kind == 2) wx = wy = wz = __builtin_nanf("");
  dst[0] = wx; dst[2] = wy; dst[4] = wz;
}

// One workgroup per 512 slots: triangle operands, then the three levels above them from LDS;
// the workgroup that finishes last merges the remaining levels (a few thousand nodes at most
// for a million triangles).
__global__ void __launch_bounds__(TREE_BLOCK)
k_pack_tree(const rtmi_triangle* __restrict__ tris, const TriPlane* __restrict__ planes, TreeDev T, d3 cam, double dir_err,
            float4* pos_plane, int2* pos_mt) {
  __shared__ NodeCone l0[TREE_BLOCK];
  __shared__ NodeCone l1[TREE_BLOCK / 8];
  __shared__ NodeCone l2[TREE_BLOCK / 64];
  __shared__ int is_last;
  const int tid = threadIdx.x;
  const int p = blockIdx.x * TREE_BLOCK + tid;
  NodeCone nc;
  nc.ax = nc.ay = nc.az = 0.0; nc.c = 1.0; nc.s = 0.0; nc.kind = 0; nc.pad = 0;
  d3 ca = rtx::mk(0.0, 0.0, 0.0), cb = ca, cc = ca;  // unit vectors from the camera to the corners of a kept triangle
  bool has_corners = false;
  if (p < T.n_slots) {
    const int ti = T.slot_tri[p];
    float e9[EDGE_FLOATS];
    for (int q = 0; q < EDGE_FLOATS; ++q) e9[q] = 0.f;
    bool keep = false;
    if (ti >= 0) {
      // t > 0 is ray-independent for a common origin (Renderables.swift:150-152): keep iff num < 0
      const TriPlane pl = planes[ti];
      double num = pl.c - rtx::dp(rtx::mk(pl.nx, pl.ny, pl.nz), cam);
      keep = num < 0;  // false for NaN (degenerate triangle): never hit in the reference either
      // the plane as the FP32 lighting reads it, filed under the triangle's list position (what a candidate carries)
      if (pos_plane) {
        pos_plane[T.slot_pos[p]] = make_float4((float)pl.nx, (float)pl.ny, (float)pl.nz, (float)num);
        pos_mt[T.slot_pos[p]] = make_int2(tris[ti].material, ti);
      }
    }
    if (keep) {
      const rtmi_triangle& t = tris[ti];
      d3 A = rtx::sub(rtx::ld3(t.v[0]), cam), B = rtx::sub(rtx::ld3(t.v[1]), cam), Cc = rtx::sub(rtx::ld3(t.v[2]), cam);
      double la = sqrt(rtx::dp(A, A)), lb = sqrt(rtx::dp(B, B)), lc = sqrt(rtx::dp(Cc, Cc));
      d3 a = rtx::divs(A, la), bb = rtx::divs(B, lb), c = rtx::divs(Cc, lc);
      d3 sum = rtx::add(rtx::add(a, bb), c);
      double ls = sqrt(rtx::dp(sum, sum));
      d3 axis = rtx::divs(sum, ls);
      double cos_t = fmin(rtx::dp(axis, a), fmin(rtx::dp(axis, bb), rtx::dp(axis, c)));
      // Narrower cones that hold the three corners, where they exist: the cone through all three (its axis is the normal
      // of the plane through the corners' unit vectors) and the cones over one side.  Whatever axis is tried, the cone
      // reaches to the corner farthest from it, so every candidate is a valid cone; the narrowest is kept.
      {
        d3 cand[4] = {rtx::cross(rtx::sub(bb, a), rtx::sub(c, a)), rtx::add(a, bb), rtx::add(bb, c), rtx::add(c, a)};
        for (int k = 0; k < 4; ++k) {
          const double l = sqrt(rtx::dp(cand[k], cand[k]));
          if (!(l > 1.0e-12)) continue;
          d3 ax = rtx::divs(cand[k], l);
          if (rtx::dp(ax, sum) < 0) ax = rtx::smul(-1.0, ax);
          const double ck = fmin(rtx::dp(ax, a), fmin(rtx::dp(ax, bb), rtx::dp(ax, c)));
          if (ck > cos_t) { cos_t = ck; axis = ax; }
        }
      }
      if (cos_t - 1.9073486328125e-06 - dir_err >= 0.0625) {  // false for NaN too
        nc.kind = 1;
        nc.ax = axis.x; nc.ay = axis.y; nc.az = axis.z;
        nc.c = cos_t;
        nc.s = sqrt(fmax(0.0, 1.0 - cos_t * cos_t));
        ca = a; cb = bb; cc = c;
        has_corners = true;
      } else {
        nc.kind = 2;
      }
      // edge planes, as in k_pack_write
      d3 M[3] = {rtx::cross(A, B), rtx::cross(B, Cc), rtx::cross(Cc, A)};
      double L = fabs(cam.x) + fabs(cam.y) + fabs(cam.z);
      double lv = 0;
      for (int q = 0; q < 3; ++q) lv = fmax(lv, fabs(t.v[q][0]) + fabs(t.v[q][1]) + fabs(t.v[q][2]));
      L += lv;
      const double eps32 = 5.9604644775390625e-08;    // 2^-24
      const double abs_term = 9.094947017729282e-13;  // 2^-40
      for (int e = 0; e < 3; ++e) {
        // (a direction that is dir_err away from the reference's moves d.M by at most dir_err * |M|_2)
        double E = (16.0 * eps32 + dir_err) * (fabs(M[e].x) + fabs(M[e].y) + fabs(M[e].z)) + abs_term * L * L;
        float mx = (float)(M[e].x / E), my = (float)(M[e].y / E), mz = (float)(M[e].z / E);
        if (!(E > 0) || !isfinite(mx) || !isfinite(my) || !isfinite(mz)) mx = my = mz = 0.0f;  // always nominate
        e9[e * 3 + 0] = mx; e9[e * 3 + 1] = my; e9[e * 3 + 2] = mz;
      }
    }
    store_cone(rec_entry(T.rec, T.rec_off[0] + (p >> 3), p & 7), &nc, dir_err);
    float* ed = T.edge + (size_t)p * EDGE_FLOATS;
    for (int q = 0; q < EDGE_FLOATS; ++q) ed[q] = e9[q];
  }
  l0[tid] = nc;
  __syncthreads();

  // Level L nodes (L = 1, 2, 3) of this workgroup: node j stands for the eight nodes of record (L-1, j); the record gets
  // its "always" mask, and the node's own cone goes into record (L, j/8) if that level exists.  Per level: the nodes'
  // axes (one work-item per node), then every work-item's least p.corner for the axis of its ancestor at that level,
  // reduced over the 8^L work-items below the node, then the cones.
  __shared__ double cm1[TREE_BLOCK / 8];
  __shared__ double cm2[TREE_BLOCK / 64];
  __shared__ double cm3[TREE_BLOCK / 64];
  NodeCone empty = nc;
  empty.kind = 0;
  const int g1 = blockIdx.x * (TREE_BLOCK / 8) + tid;
  const int g2 = blockIdx.x * (TREE_BLOCK / 64) + tid;
  unsigned always = 0;
  if (tid < TREE_BLOCK / 8) {
    NodeCone m = empty;
    if (1 <= T.ltop + 1 && g1 < T.n_nodes[1]) m = node_axis(&l0[tid * 8], 8, &always);
    l1[tid] = m;
  }
  __syncthreads();
  {
    double v = corner_min(l1[tid >> 3], has_corners, ca, cb, cc);
    for (int sh = 1; sh < 8; sh <<= 1) v = fmin(v, __shfl_xor(v, sh));
    if ((tid & 7) == 0) cm1[tid >> 3] = v;
  }
  __syncthreads();
  if (tid < TREE_BLOCK / 8 && 1 <= T.ltop + 1 && g1 < T.n_nodes[1]) {
    NodeCone m = l1[tid];
    node_finish(&m, cm1[tid]);
    if (1 <= T.ltop) store_cone(rec_entry(T.rec, T.rec_off[1] + (g1 >> 3), g1 & 7), &m, dir_err);
    reinterpret_cast<unsigned*>(T.rec)[(size_t)(T.rec_off[0] + g1) * REC_DWORDS + REC_ALWAYS] = always;
    l1[tid] = m;
  }
  __syncthreads();
  if (tid < TREE_BLOCK / 64) {
    NodeCone m = empty;
    if (2 <= T.ltop + 1 && g2 < T.n_nodes[2]) m = node_axis(&l1[tid * 8], 8, &always);
    l2[tid] = m;
  }
  __syncthreads();
  {
    double v = corner_min(l2[tid >> 6], has_corners, ca, cb, cc);
    for (int sh = 1; sh < 64; sh <<= 1) v = fmin(v, __shfl_xor(v, sh));
    if ((tid & 63) == 0) cm2[tid >> 6] = v;
  }
  __syncthreads();
  if (tid < TREE_BLOCK / 64 && 2 <= T.ltop + 1 && g2 < T.n_nodes[2]) {
    NodeCone m = l2[tid];
    node_finish(&m, cm2[tid]);
    if (2 <= T.ltop) store_cone(rec_entry(T.rec, T.rec_off[2] + (g2 >> 3), g2 & 7), &m, dir_err);
    reinterpret_cast<unsigned*>(T.rec)[(size_t)(T.rec_off[1] + g2) * REC_DWORDS + REC_ALWAYS] = always;
    l2[tid] = m;
  }
  __syncthreads();
  if (T.ltop + 1 < 3) return;
  __shared__ NodeCone l3;
  const int g3 = blockIdx.x;
  if (tid == 0) {
    NodeCone m = empty;
    if (g3 < T.n_nodes[3]) m = node_axis(l2, 8, &always);
    l3 = m;
  }
  __syncthreads();
  {
    double v = corner_min(l3, has_corners, ca, cb, cc);
    for (int sh = 1; sh < 64; sh <<= 1) v = fmin(v, __shfl_xor(v, sh));
    if ((tid & 63) == 0) cm3[tid >> 6] = v;
  }
  __syncthreads();
  if (tid == 0) {
    if (g3 < T.n_nodes[3]) {
      NodeCone m = l3;
      double v = cm3[0];
      for (int w = 1; w < TREE_BLOCK / 64; ++w) v = fmin(v, cm3[w]);
      node_finish(&m, v);
      if (3 <= T.ltop) store_cone(rec_entry(T.rec, T.rec_off[3] + (g3 >> 3), g3 & 7), &m, dir_err);
      reinterpret_cast<unsigned*>(T.rec)[(size_t)(T.rec_off[2] + g3) * REC_DWORDS + REC_ALWAYS] = always;
      T.nodes[T.node_off[3] + g3] = m;
    }
    __threadfence();
    is_last = atomicAdd(T.ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last || T.ltop + 1 < 4) {
    if (is_last && tid == 0) *T.ticket = 0;
    return;
  }
  __threadfence();
  for (int L = 4; L <= T.ltop + 1; ++L) {
    for (int j = tid; j < T.n_nodes[L]; j += TREE_BLOCK) {
      NodeCone ch[8];
      int nv = min(8, T.n_nodes[L - 1] - j * 8);
      for (int k = 0; k < 8; ++k)
        if (k < nv) ch[k] = T.nodes[T.node_off[L - 1] + j * 8 + k];
      unsigned always;
      NodeCone m = merge_cones(ch, nv, &always);
      if (L <= T.ltop) store_cone(rec_entry(T.rec, T.rec_off[L] + (j >> 3), j & 7), &m, dir_err);
      reinterpret_cast<unsigned*>(T.rec)[(size_t)(T.rec_off[L - 1] + j) * REC_DWORDS + REC_ALWAYS] = always;
      T.nodes[T.node_off[L] + j] = m;
    }
    __threadfence();
    __syncthreads();
  }
  if (tid == 0) *T.ticket = 0;
}
