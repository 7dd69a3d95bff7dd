// TEST INFRASTRUCTURE (never part of the product path): a CPU model of the stack-free first-leaf search of
// k_raycast (supereight_amd/csrc/se_kernels.h, se_first_leaf_lite) checked ray by ray against the oracle's restatement
// of se::ray_iterator (se_core/include/se/ray_iterator.hpp:53-226, oracle/se_oracle.cpp RayIterator).
//
// What the model claims.  The reference iterator carries, beside (pos, scale, t_min), a stack of (parent, t_max) per
// scale and the value h.  For a ray that is *regular* at set-up (t_min < h: it enters the volume before it leaves it,
// nothing is NaN) those carry no information:
//   * parent: the node whose children are the cells of scale s is a function of pos (its heap code is the current
//     code shifted right by three bits per level popped);
//   * t_max = min(far / dim, root exit, tc_max of every ancestor cell on the path).  Every ancestor's tc_max is >= the
//     tc_max of any cell inside it (t = pos * t_coef - t_bias is monotone in pos), and t_min only takes values tc_max of
//     cells inside the current ancestors -- so `t_min <= t_max` is `t_min <= min(far / dim, root exit)` unless the ray
//     descended from a cell whose own tc_max was already below t_min.  That can happen: the child slot is chosen by
//     t_center = half * t_coef + t_corner, a different rounding of the same plane than the child's own t_corner, so a
//     ray that passes within an ulp of a cell edge may be placed in a child it has, by t_corner, already left.  The
//     model FLAGS exactly that event (descent with tc_max < t_min) and the kernel re-runs a flagged ray through the
//     full iterator; so does a ray that is not regular at set-up;
//   * h only decides whether a stack slot is (re)written, never what is read back (a slot that is read was written by
//     the first descent below the same parent).
// The test runs every pixel of a few frames through both and requires bit-identical t_min (leaf found or not) for all
// rays that are neither flagged nor irregular, and reports how many are.
//
// r05, beam start (se_beam_start / the t_start argument of se_first_leaf_lite): the 64 rays of an 8x8 pixel tile enter the tree at
// t_start = the distance up to which a dilated coarse occupancy bitmap shows the whole beam clear, instead of at the near plane.  The model
// restates that pre-pass in the kernel's own float arithmetic (beam_start below: same sample points, same bound, same 27-neighbourhood
// dilation of the allocated blocks) and runs `lite` from there; fl_compare(..., beam = 1) then requires the same bit-identical t_min /
// found decision against the iterator started at the near plane, and reports how many trips the jump saves.
#include "../../oracle/se_oracle.cpp"

namespace fl {

struct Result { float t_min; int found; int flagged; int irregular; int trips; float t_lim; };
static thread_local int* g_adv_hist = nullptr;   // (experiments: advances per scale, descents at [32 + scale])

// the node whose children are the cells of `scale`, found from pos alone (what `code >> 3 * levels` is on the device)
template <typename VT> static Node<VT>* node_at(const Octree<VT>& m, V3f pos, int scale, int om) {
  Node<VT>* n = m.root_;
  for (int s = CAST_STACK_DEPTH - 1; s > scale && n; --s) {
    const int idx = ((f2i(pos.x) >> s) & 1) | (((f2i(pos.y) >> s) & 1) << 1) | (((f2i(pos.z) >> s) & 1) << 2);
    n = n->child(idx ^ om);
  }
  return n;
}

template <typename VT> static Result lite(const Octree<VT>& m, V3f origin, V3f direction, float nearP, float farP, float t_start = 0.f) {
  Result r = {0.f, 0, 0, 0, 0, 0.f};
  V3f pos = {1.f, 1.f, 1.f};
  int scale = CAST_STACK_DEPTH - 1;
  float scale_exp2 = 0.5f;
  const int min_scale = CAST_STACK_DEPTH - (int)std::log2((double)(m.size_ / BLOCK_SIDE));
  const float epsilon = exp2f(-(float)std::log2((double)m.size_));
  V3f d;
  d.x = fabsf(direction.x) < epsilon ? copysignf(epsilon, direction.x) : direction.x;
  d.y = fabsf(direction.y) < epsilon ? copysignf(epsilon, direction.y) : direction.y;
  d.z = fabsf(direction.z) < epsilon ? copysignf(epsilon, direction.z) : direction.z;
  const V3f so = origin / m.dim_ + V3f{1.f, 1.f, 1.f};
  const V3f tc = -1.f * V3f{1.f / fabsf(d.x), 1.f / fabsf(d.y), 1.f / fabsf(d.z)};
  V3f tb = cwise(tc, so);
  int om = 0;  // octant_mask ^ 7
  if (d.x > 0.f) om ^= 1, tb.x = 3.f * tc.x - tb.x;
  if (d.y > 0.f) om ^= 2, tb.y = 3.f * tc.y - tb.y;
  if (d.z > 0.f) om ^= 4, tb.z = 3.f * tc.z - tb.z;
  float t_min = fmaxf(fmaxf(2.f * tc.x - tb.x, 2.f * tc.y - tb.y), 2.f * tc.z - tb.z);
  const float h0 = fminf(fminf(tc.x - tb.x, tc.y - tb.y), tc.z - tb.z);
  t_min = fmaxf(t_min, nearP / m.dim_);
  const float t_lim = fminf(h0, farP / m.dim_);   // t_max_init: the only t_max the model knows
  if (!(t_min < h0)) { r.irregular = 1; r.t_min = t_min; return r; }
  const bool jumped = t_start > t_min;
  if (jumped) t_min = t_start;
  if (1.5f * tc.x - tb.x > t_min) pos.x = 1.5f;
  if (1.5f * tc.y - tb.y > t_min) pos.y = 1.5f;
  if (1.5f * tc.z - tb.z > t_min) pos.z = 1.5f;
  Node<VT>* parent = m.root_;
  while (scale < CAST_STACK_DEPTH) {
    ++r.trips;
    const V3f t_corner = cwise(pos, tc) - tb;
    const float tc_max = fminf(fminf(t_corner.x, t_corner.y), t_corner.z);
    const int idx = ((f2i(pos.x) >> scale) & 1) | (((f2i(pos.y) >> scale) & 1) << 1) | (((f2i(pos.z) >> scale) & 1) << 2);
    if (!parent) { r.flagged = 2; break; }   // (would be a bug of the model: counted, never expected)
    Node<VT>* child = parent->child(idx ^ om);
    if (scale == min_scale && child) { r.found = 1; break; }
    if (child && t_min <= t_lim) {
      if (tc_max < t_min) { r.flagged = 1; break; }
      const float half = scale_exp2 * 0.5f;
      const V3f t_center = half * tc + t_corner;
      if (g_adv_hist) ++g_adv_hist[32 + scale];
      parent = child;
      --scale;
      scale_exp2 = half;
      if (t_center.x > t_min) pos.x += half;
      if (t_center.y > t_min) pos.y += half;
      if (t_center.z > t_min) pos.z += half;
      continue;
    }
    // advance_ray without idx_: "leaves the parent" = a bit above `scale` changed
    if (g_adv_hist) ++g_adv_hist[scale];
    const V3f old = pos;
    if (t_corner.x <= tc_max) pos.x -= scale_exp2;
    if (t_corner.y <= tc_max) pos.y -= scale_exp2;
    if (t_corner.z <= tc_max) pos.z -= scale_exp2;
    t_min = tc_max;
    const unsigned diff = (unsigned)(f2i(old.x) ^ f2i(pos.x)) | (unsigned)(f2i(old.y) ^ f2i(pos.y)) | (unsigned)(f2i(old.z) ^ f2i(pos.z));
    if (diff > (1u << scale)) {
      scale = 31 - __builtin_clz(diff);
      scale_exp2 = i2f((scale - CAST_STACK_DEPTH + 127) << 23);
      if (scale < CAST_STACK_DEPTH) {
        const int keep = (int)(0xFFFFFFFFu << scale);
        pos.x = i2f(f2i(pos.x) & keep); pos.y = i2f(f2i(pos.y) & keep); pos.z = i2f(f2i(pos.z) & keep);
        parent = node_at(m, pos, scale, om);
      }
    }
  }
  if (jumped && t_min == t_start && !r.flagged) r.flagged = 1;   // never advanced: would return t_start as an entry time -> handed back
  r.t_min = t_min;
  return r;
}

// r06 experiment: the first leaf by a FLAT walk over leaf-size cells (no tree, no descents, no pops): the iterator's own plane times t = pos * t_coef - t_bias
// decide every step, so the time at which the walk enters the first allocated leaf is the iterator's t_min bit for bit -- the plane through which a ray
// enters a cell is the same plane whatever cells came before.  What the tree adds to the reference's answer is only its far-plane rule: a node is descended
// into while t_min <= t_max, leaves of a node already entered are returned whatever their distance -- i.e. a leaf counts iff its PARENT cell (level
// leaf - 1) was entered at t <= t_lim.  Flags (handed back to the full iterator): a cell entered after its own exit time (the rounding artefact `lite` flags
// at descents), and, if `flag_ties`, a step that crosses two planes at the same float time.
// Outcome (se_kernels.h, in front of se_first_leaf_lite; profiles/r06m_flat_walk_ab.log): not adopted.  On the device it was no faster (512^3 +-0, 2048^3 and the
// stress streams slower), and "bit for bit" above holds only away from near-ties: the iterator places a ray inside a cell it descends into by t_center
// (half * t_coef + t_corner), the walk orders the same planes by t_corner of the leaf cells, and where two crossings fall within a few ulp the two disagree about
// which plane the ray entered its block through -- stress stream, 640x480, 512^3, frame 11, pixel (224, 222): t_min 0.28818703 instead of 0.28818679.  About one
// ray in 10^6-10^7; the streams below did not contain one.  Kept as the record of the experiment; nothing on the product path uses it.
template <typename VT> static Result flat(const Octree<VT>& m, V3f origin, V3f direction, float nearP, float farP, float t_start, bool flag_ties) {
  Result r = {0.f, 0, 0, 0, 0, 0.f};
  const int min_scale = CAST_STACK_DEPTH - (int)std::log2((double)(m.size_ / BLOCK_SIDE));
  const float epsilon = exp2f(-(float)std::log2((double)m.size_));
  V3f d;
  d.x = fabsf(direction.x) < epsilon ? copysignf(epsilon, direction.x) : direction.x;
  d.y = fabsf(direction.y) < epsilon ? copysignf(epsilon, direction.y) : direction.y;
  d.z = fabsf(direction.z) < epsilon ? copysignf(epsilon, direction.z) : direction.z;
  const V3f so = origin / m.dim_ + V3f{1.f, 1.f, 1.f};
  const V3f tc = -1.f * V3f{1.f / fabsf(d.x), 1.f / fabsf(d.y), 1.f / fabsf(d.z)};
  V3f tb = cwise(tc, so);
  int om = 0;
  if (d.x > 0.f) om ^= 1, tb.x = 3.f * tc.x - tb.x;
  if (d.y > 0.f) om ^= 2, tb.y = 3.f * tc.y - tb.y;
  if (d.z > 0.f) om ^= 4, tb.z = 3.f * tc.z - tb.z;
  float t_min = fmaxf(fmaxf(2.f * tc.x - tb.x, 2.f * tc.y - tb.y), 2.f * tc.z - tb.z);
  const float h0 = fminf(fminf(tc.x - tb.x, tc.y - tb.y), tc.z - tb.z);
  t_min = fmaxf(t_min, nearP / m.dim_);
  const float t_lim = fminf(h0, farP / m.dim_);
  r.t_lim = t_lim;
  if (!(t_min < h0)) { r.irregular = 1; r.t_min = t_min; return r; }
  const bool jumped = t_start > t_min;
  if (jumped) t_min = t_start;
  if (!(t_min < h0)) { r.t_min = t_min; return r; }   // the beam is clear beyond this ray's own exit from the cube: nothing to find (t_min >= t_lim)
  // the leaf-size cell the ray is in at t_min: the iterator's own child choices, level by level (no tree needed for them)
  V3f pos = {1.f, 1.f, 1.f};
  if (1.5f * tc.x - tb.x > t_min) pos.x = 1.5f;
  if (1.5f * tc.y - tb.y > t_min) pos.y = 1.5f;
  if (1.5f * tc.z - tb.z > t_min) pos.z = 1.5f;
  float e = 0.5f;
  for (int s = CAST_STACK_DEPTH - 1; s > min_scale; --s) {
    const V3f t_corner = cwise(pos, tc) - tb;
    const float tcm = fminf(fminf(t_corner.x, t_corner.y), t_corner.z);
    if (tcm < t_min) { r.flagged = 1; r.t_min = t_min; return r; }
    const float half = e * 0.5f;
    const V3f t_center = half * tc + t_corner;
    if (t_center.x > t_min) pos.x += half;
    if (t_center.y > t_min) pos.y += half;
    if (t_center.z > t_min) pos.z += half;
    e = half;
  }
  float t_parent = t_min;   // when the current leaf-parent cell was entered
  for (;;) {
    ++r.trips;
    Node<VT>* lp = node_at(m, pos, min_scale, om);
    const int idx = ((f2i(pos.x) >> min_scale) & 1) | (((f2i(pos.y) >> min_scale) & 1) << 1) | (((f2i(pos.z) >> min_scale) & 1) << 2);
    if (lp && lp->child(idx ^ om) && t_parent <= t_lim) { r.found = 1; break; }
    const V3f t_corner = cwise(pos, tc) - tb;
    const float tcm = fminf(fminf(t_corner.x, t_corner.y), t_corner.z);
    if (tcm < t_min) { r.flagged = 1; break; }
    const V3f old = pos;
    int nax = 0;
    if (t_corner.x <= tcm) pos.x -= e, ++nax;
    if (t_corner.y <= tcm) pos.y -= e, ++nax;
    if (t_corner.z <= tcm) pos.z -= e, ++nax;
    t_min = tcm;
    if (flag_ties && nax > 1) { r.flagged = 1; break; }
    const unsigned diff = (unsigned)(f2i(old.x) ^ f2i(pos.x)) | (unsigned)(f2i(old.y) ^ f2i(pos.y)) | (unsigned)(f2i(old.z) ^ f2i(pos.z));
    if (diff >= (1u << CAST_STACK_DEPTH)) break;                 // left the cube: nothing found
    if (diff >= (1u << (min_scale + 1))) {                       // a new leaf-parent cell
      t_parent = t_min;
      if (t_min > t_lim) break;                                  // the iterator descends into no further node
    }
    if (r.trips > 100000) { r.flagged = 2; break; }
  }
  if (jumped && t_min == t_start && !r.flagged) r.flagged = 1;
  r.t_min = t_min;
  return r;
}

// the dilated coarse bitmap of se_device.h (DevMap::cbits), built from the oracle's block pool
template <typename VT> static std::vector<uint32_t> coarse_bits(const Octree<VT>& m, int C) {
  const int leaf_level = m.max_level_ - 3, sh = leaf_level - C, n = 1 << C;
  std::vector<uint32_t> bits(std::max<size_t>(1, ((size_t)1 << (3 * C)) / 32), 0u);
  for (size_t i = 0; i < m.block_buffer_.size(); ++i) {
    const V3i c = m.block_buffer_[i]->coordinates_;
    const int cx = (c.x >> 3) >> sh, cy = (c.y >> 3) >> sh, cz = (c.z >> 3) >> sh;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int ux = cx + dx, uy = cy + dy, uz = cz + dz;
          if ((unsigned)ux >= (unsigned)n || (unsigned)uy >= (unsigned)n || (unsigned)uz >= (unsigned)n) continue;
          const uint32_t idx = ((uint32_t)uz << (2 * C)) | ((uint32_t)uy << C) | (uint32_t)ux;
          bits[idx >> 5] |= 1u << (idx & 31u);
        }
  }
  return bits;
}
// se_beam_start of se_kernels.h for the tile whose first pixel is (x0, y0): the 64 "lanes" are the tile's pixels
template <typename VT> static float beam_start(const Octree<VT>& m, const std::vector<uint32_t>& cbits, int C, const M4& view, int x0, int y0, float nearP, float farP,
                                               const std::vector<uint32_t>* fbits = nullptr, int Fl = 0) {
  const V3f org = {view.m[0][3], view.m[1][3], view.m[2][3]};
  const V3f dc = normalized(mul3(top3(view), {(float)x0 + 3.5f, (float)y0 + 3.5f, 1.f}));
  float dev = 0.f;
  for (int l = 0; l < 64; ++l) {
    const V3f dir = normalized(mul3(top3(view), {(float)(x0 + (l & 7)), (float)(y0 + (l >> 3)), 1.f}));
    const V3f d = dir - dc;
    dev = fmaxf(dev, sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z));
  }
  const float epsilon = exp2f(-(float)std::log2((double)m.size_));
  const float rad = dev * 1.05f + epsilon;
  const float cell = m.dim_ / (float)(1 << C), inv_cell = (float)(1 << C) / m.dim_, inv_dim = 1.f / m.dim_;
  const float dt = std::max(0.5f * cell, (farP - nearP) / 64.f);
  int j = 64;
  for (int l = 0; l < 64; ++l) {
    const float ti = nearP + (float)l * dt;
    const V3f p = org + dc * ti;
    const int cx = (int)floorf(p.x * inv_cell), cy = (int)floorf(p.y * inv_cell), cz = (int)floorf(p.z * inv_cell);
    // (a sample in the one-cell shell around the volume takes the dilated bit of the boundary cell it touches, as the kernel does)
    const int nC = 1 << C;
    auto cl = [](int v, int n) { return std::min(std::max(v, 0), n - 1); };
    const bool in = (uint32_t)(cx + 1) <= (uint32_t)nC && (uint32_t)(cy + 1) <= (uint32_t)nC && (uint32_t)(cz + 1) <= (uint32_t)nC;
    const uint32_t idx = in ? (((uint32_t)cl(cz, nC) << (2 * C)) | ((uint32_t)cl(cy, nC) << C) | (uint32_t)cl(cx, nC)) : 0u;
    const bool occupied = in && ((cbits[idx >> 5] >> (idx & 31u)) & 1u);
    const bool clear = !occupied && ((ti + 0.5f * dt) * rad + 0.5f * dt <= 0.9f * cell);
    if (!clear) { j = l; break; }
  }
  if (j < 1) return 0.f;
  float t_safe = nearP + ((float)j - 0.5f) * dt;
  if (fbits) {
    // stage 2: the same test on the fine grid (level Fl, dilated by one fine cell), 64 samples from t_safe on
    const float cellf = m.dim_ / (float)(1 << Fl), inv_cellf = (float)(1 << Fl) / m.dim_;
    const float dt2 = 0.4f * cellf;
    int j2 = 64;
    for (int l = 0; l < 64; ++l) {
      const float ti = t_safe + ((float)l + 0.5f) * dt2;
      const V3f p = org + dc * ti;
      const int cx = (int)floorf(p.x * inv_cellf), cy = (int)floorf(p.y * inv_cellf), cz = (int)floorf(p.z * inv_cellf);
      const int nF = 1 << Fl;
      auto cl = [](int v, int n) { return std::min(std::max(v, 0), n - 1); };
      const bool in = (uint32_t)(cx + 1) <= (uint32_t)nF && (uint32_t)(cy + 1) <= (uint32_t)nF && (uint32_t)(cz + 1) <= (uint32_t)nF;
      const uint32_t idx = in ? (((uint32_t)cl(cz, nF) << (2 * Fl)) | ((uint32_t)cl(cy, nF) << Fl) | (uint32_t)cl(cx, nF)) : 0u;
      const bool occupied = in && (((*fbits)[idx >> 5] >> (idx & 31u)) & 1u);
      const bool clear = !occupied && ((ti + 0.5f * dt2) * rad + 0.5f * dt2 <= 0.9f * cellf);
      if (!clear) { j2 = l; break; }
    }
    t_safe += (float)j2 * dt2;
  }
  // never beyond the far plane: the iterator descends into a node only while t_min <= far / dim, but returns leaves of a node it is already in
  // whatever their distance -- a start behind the far plane would never descend and miss those
  return fminf(t_safe, farP) * inv_dim;
}

template <typename VT> static void compare(Pipeline<VT>* p, const float* pose_cm, const float* k, int64_t* out, int32_t* first_bad, int beam) {
  const bool use_flat = (beam & 0x100) != 0, flat_ties = (beam & 0x200) != 0;   // (r06 experiment: the flat walk instead of `lite`)
  beam &= 0xFF;
  const M4 view = mul(from_colmajor(pose_cm), inverse_camera_matrix(k));
  const Octree<VT>& oct = p->oct;
  const int C = std::min(oct.max_level_ - 3, 5);
  std::vector<uint32_t> cbits;
  std::vector<float> tile_start;
  const int tiles_x = (p->W + 7) / 8, tiles_y = (p->H + 7) / 8;
  if (beam) {
    cbits = coarse_bits(oct, C);
    // the second stage's grid: level min(leaf, 6) (DevMap::flevel, se_device.h); bits 4.. of `beam` override it (experiments)
    const int Fl = (beam >> 4) ? (beam >> 4) : std::min(oct.max_level_ - 3, 6);
    beam &= 15;
    std::vector<uint32_t> fbits;
    if (beam >= 2 && Fl > C) fbits = coarse_bits(oct, Fl);
    tile_start.resize((size_t)tiles_x * tiles_y);
    for (int ty = 0; ty < tiles_y; ++ty)
      for (int tx = 0; tx < tiles_x; ++tx)
        tile_start[(size_t)ty * tiles_x + tx] = beam_start(oct, cbits, C, view, tx * 8, ty * 8, nearPlane, farPlane, fbits.empty() ? nullptr : &fbits, Fl);
  }
  int64_t rays = 0, irregular = 0, flagged = 0, mismatch = 0, found = 0, trips_ref = 0, trips_lite = 0, model_bug = 0;
  int bad_x = -1, bad_y = -1;
#pragma omp parallel for reduction(+ : rays, irregular, flagged, mismatch, found, trips_ref, trips_lite, model_bug)
  for (int y = 0; y < p->H; ++y)
    for (int x = 0; x < p->W; ++x) {
      const V3f dir = normalized(mul3(top3(view), {(float)x, (float)y, 1.f}));
      const V3f transl = {view.m[0][3], view.m[1][3], view.m[2][3]};
      g_ray_iter = 0;
      RayIterator<VT> ray(oct, transl, dir, nearPlane, farPlane);
      const bool ref_found = ray.next() != nullptr;
      const float ref_t = ray.t_min_;
      trips_ref += g_ray_iter;
      const float ts = beam ? tile_start[(size_t)(y / 8) * tiles_x + x / 8] : 0.f;
      const Result r = use_flat ? flat(oct, transl, dir, nearPlane, farPlane, ts, flat_ties) : lite(oct, transl, dir, nearPlane, farPlane, ts);
      ++rays;
      trips_lite += r.trips;
      if (r.irregular) { ++irregular; continue; }
      if (r.flagged == 2) { ++model_bug; continue; }
      if (r.flagged) { ++flagged; continue; }
      found += r.found;
      // (the flat walk stops at the far plane: a ray that finds nothing only has to agree on "t_min is not in front of t_max", which is all raycastKernel
      // looks at then -- rendering.cpp:60-62: raycast(...) is entered with tnear = t_min, tfar = tmax and returns at once)
      const bool both_none = use_flat && !r.found && !ref_found && r.t_min >= r.t_lim && ref_t >= r.t_lim;
      if (!both_none && (f2i(r.t_min) != f2i(ref_t) || (r.found != 0) != ref_found)) {
        ++mismatch;
#pragma omp critical
        { bad_x = x; bad_y = y; }
      }
    }
  out[0] = rays; out[1] = irregular; out[2] = flagged; out[3] = mismatch; out[4] = found; out[5] = trips_ref; out[6] = trips_lite; out[7] = model_bug;
  first_bad[0] = bad_x; first_bad[1] = bad_y;
}

}  // namespace fl

extern "C" void fl_compare(void* pipe, const float* pose_cm, const float* k, int64_t* out, int32_t* first_bad, int beam) {
  PipelineBase* b = (PipelineBase*)pipe;
  if (auto* s = dynamic_cast<Pipeline<SDFv>*>(b)) fl::compare(s, pose_cm, k, out, first_bad, beam);
  else if (auto* o = dynamic_cast<Pipeline<OFv>*>(b)) fl::compare(o, pose_cm, k, out, first_bad, beam);
}

// debug: one pixel
// experiments: trips of every ray (lite, with the beam start) and, summed over the rays with >= min_trips trips, advances per scale [0..31] / descents [32..63]
extern "C" void fl_trip_map(void* pipe, const float* pose_cm, const float* k, int beam, int min_trips, int32_t* trips, int64_t* hist) {
  auto* p = dynamic_cast<Pipeline<SDFv>*>((PipelineBase*)pipe);
  const M4 view = mul(from_colmajor(pose_cm), inverse_camera_matrix(k));
  const Octree<SDFv>& oct = p->oct;
  const int C = std::min(oct.max_level_ - 3, 5), Fl = std::min(oct.max_level_ - 3, 6);
  auto cb = fl::coarse_bits(oct, C);
  std::vector<uint32_t> fb;
  if (beam >= 2 && Fl > C) fb = fl::coarse_bits(oct, Fl);
  const int tiles_x = (p->W + 7) / 8, tiles_y = (p->H + 7) / 8;
  std::vector<float> ts((size_t)tiles_x * tiles_y, 0.f);
  if (beam)
    for (int ty = 0; ty < tiles_y; ++ty)
      for (int tx = 0; tx < tiles_x; ++tx) ts[(size_t)ty * tiles_x + tx] = fl::beam_start(oct, cb, C, view, tx * 8, ty * 8, nearPlane, farPlane, fb.empty() ? nullptr : &fb, Fl);
  for (int i = 0; i < 64; ++i) hist[i] = 0;
  for (int y = 0; y < p->H; ++y)
    for (int x = 0; x < p->W; ++x) {
      const V3f dir = normalized(mul3(top3(view), {(float)x, (float)y, 1.f}));
      const V3f transl = {view.m[0][3], view.m[1][3], view.m[2][3]};
      int h[64] = {0};
      fl::g_adv_hist = h;
      const fl::Result r = fl::lite(oct, transl, dir, nearPlane, farPlane, ts[(size_t)(y / 8) * tiles_x + x / 8]);
      fl::g_adv_hist = nullptr;
      trips[(size_t)y * p->W + x] = r.trips;
      if (r.trips >= min_trips) for (int i = 0; i < 64; ++i) hist[i] += h[i];
    }
}
extern "C" void fl_debug(void* pipe, const float* pose_cm, const float* k, int x, int y, int beam, double* out) {
  auto* p = dynamic_cast<Pipeline<SDFv>*>((PipelineBase*)pipe);
  const M4 view = mul(from_colmajor(pose_cm), inverse_camera_matrix(k));
  const Octree<SDFv>& oct = p->oct;
  const int C = std::min(oct.max_level_ - 3, 5), Fl = std::min(oct.max_level_ - 3, 6);
  auto cb = fl::coarse_bits(oct, C), fb = fl::coarse_bits(oct, Fl);
  const float ts = fl::beam_start(oct, cb, C, view, (x / 8) * 8, (y / 8) * 8, nearPlane, farPlane, beam >= 2 ? &fb : nullptr, Fl);
  const V3f dir = normalized(mul3(top3(view), {(float)x, (float)y, 1.f}));
  const V3f transl = {view.m[0][3], view.m[1][3], view.m[2][3]};
  RayIterator<SDFv> ray(oct, transl, dir, nearPlane, farPlane);
  const bool ref_found = ray.next() != nullptr;
  const fl::Result a = fl::lite(oct, transl, dir, nearPlane, farPlane, 0.f), b = fl::lite(oct, transl, dir, nearPlane, farPlane, ts);
  { const fl::Result f = fl::flat(oct, transl, dir, nearPlane, farPlane, ts, false);
    out[10] = f.t_min * oct.dim_; out[11] = f.found; out[12] = f.flagged; out[13] = f.trips; out[14] = ray.t_min_; out[15] = f.t_min; }
  out[0] = ts * oct.dim_; out[1] = ray.t_min_ * oct.dim_; out[2] = ref_found; out[3] = a.t_min * oct.dim_; out[4] = a.found; out[5] = b.t_min * oct.dim_; out[6] = b.found; out[7] = b.flagged; out[8] = a.trips; out[9] = b.trips;
}
