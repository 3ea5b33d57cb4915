// Device-side pieces of the segmented projection engine, shared by the single-problem launches (ccqp_segs.hip) and
// the lock-step batch form (ccqp_bsegs.hip): chunk geometry, the fused input u = a*x + b*y, the Sphere / Cone rule
// for a block whose norm is known, the one-thread-per-small-block routine and its emitters.  One definition => the
// batched solves have the bits of the single-problem ones.
#pragma once
#include "ccqp_bodies.h"

__host__ __device__ static inline void segs_geometry(int64_t n, int64_t* chunk, int* nblk) {
  int64_t c = CCQP_RED_CHUNK;
  if ((n + c - 1) / c > CCQP_RED_MAXBLK) {
    c = (n + CCQP_RED_MAXBLK - 1) / CCQP_RED_MAXBLK;
    c = (c + CCQP_BLOCK - 1) / CCQP_BLOCK * CCQP_BLOCK;
  }
  int64_t nb = n > 0 ? (n + c - 1) / c : 1;
  *chunk = c;
  *nblk = (int)nb;
}

#ifdef __HIPCC__
struct segs_in {          // u_i = a*x_i + b*y_i (y == NULL: u = x), block-local pointers
  const double* x;
  const double* y;
  double a, b;
  __device__ __forceinline__ double operator()(int64_t i) const {
    double u = x[i];
    if (y) {
      const double t1 = a * u;
      const double t2 = b * y[i];
      u = t1 + t2;
    }
    return u;
  }
};

// a block's |.|^2 from its chunk sums: the single partial, or the index-order sum ccqp_reduce_finish performs
__device__ __forceinline__ double segs_finish(const double* part, int nblk, double* lds4, double* bcast) {
  double s;
  if (nblk == 1) {
    s = part[0];
  } else {
    const double t = partial_sum_body(part, nblk, lds4);
    if (threadIdx.x == 0) *bcast = t;
    __syncthreads();
    s = *bcast;
    __syncthreads();
  }
  return s;
}

// P of one element of a LARGE Sphere / Cone block, r = |u| known (normscale_kernel, solution_spaces.py:430-435,
// 484-492)
struct segs_norm_rule {
  int kind, branch;     // branch: 0 copy, 1 zeros, 2 scale
  double r, radius, mu, s;
  __device__ __forceinline__ void setup(const ccqp_seg& sg, double norm2, double zlast) {
    kind = sg.kind;
    r = sqrt(norm2);
    if (kind == CCQP_OP_SPHERE) {
      radius = sg.lb_s;
      branch = (r > radius) ? 2 : 0;
    } else {
      mu = sg.lb_s;
      if (mu * zlast >= r) branch = 0;
      else if ((-zlast) / mu >= r) branch = 1;
      else {
        branch = 2;
        const double num = zlast + mu * r;
        s = num / sg.ub_s;
      }
    }
  }
  __device__ __forceinline__ double operator()(double u, bool is_last) const {
    if (branch == 0) return u;
    if (branch == 1) return 0.0;
    if (kind == CCQP_OP_SPHERE) {
      const double t = radius * u;
      return t / r;
    }
    const double e = is_last ? -mu : u / r;
    return s * e;
  }
};

// one SMALL block, sequentially by one thread; emit(i, u_i, p_i) receives every element's input and projection
template <class Emit>
__device__ __forceinline__ void segs_small(const ccqp_seg& sg, const segs_in& u, Emit& emit) {
  const int64_t d = sg.dim;
  if (sg.kind <= CCQP_OP_BOX) {
    for (int64_t i = 0; i < d; ++i) {
      const double v = u(i);
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      emit(i, v, ccqp_proj_elem(sg.kind, sg.flags, v, lb, ub));
    }
    return;
  }
  double n2 = 0.0;
  for (int64_t i = 0; i < d; ++i) {
    const double v = u(i);
    n2 = fma(v, v, n2);
  }
  segs_norm_rule rule;
  rule.setup(sg, n2, u(d - 1));
  for (int64_t i = 0; i < d; ++i) {
    const double v = u(i);
    emit(i, v, rule(v, i == d - 1));
  }
}

struct emit_store {       // out = P(u)  |  out = P(u) - z
  double* out;
  const double* z;
  __device__ __forceinline__ void operator()(int64_t i, double, double p) { out[i] = z ? p - z[i] : p; }
};

struct emit_residual {    // acc += (c*(x - P(u)))^2, u = x - gd*g
  const double* x;
  double c, acc;
  __device__ __forceinline__ void operator()(int64_t i, double, double p) {
    const double d = x[i] - p;
    const double v = c * d;
    acc = fma(v, v, acc);
  }
};

struct emit_notclose {    // acc += 1 where not isclose(u, P(u))  (solvers.py:1115: `np.all(np.isclose(yf, P(yf)))` <=> 0)
  double acc;
  __device__ __forceinline__ void operator()(int64_t, double u, double p) { acc += ccqp_isclose(u, p) ? 0.0 : 1.0; }
};

// chunk sum of |u|^2 for one norm work item (same thread -> element map and fma chain as dot_body); valid in thread 0
__device__ __forceinline__ double segs_norm_item(const ccqp_seg* __restrict__ table, const ccqp_segwork& w, double a,
                                                 const double* x, double b, const double* y, double* lds4) {
  const ccqp_seg sg = table[w.seg];
  int64_t chunk;
  int nblk;
  segs_geometry(sg.dim, &chunk, &nblk);
  const int64_t lo = (int64_t)w.chunk * chunk;
  int64_t hi = lo + chunk;
  if (hi > sg.dim) hi = sg.dim;
  const segs_in u = {x + sg.offset, y ? y + sg.offset : nullptr, a, b};
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    const double v = u(i);
    acc = fma(v, v, acc);
  }
  return ccqp_block_sum(acc, lds4);
}

// One work item of the apply pass.  EPI 0: out = P(u) (- z), returns 0;  EPI 2: returns this thread's count of entries
// with not isclose(u, P(u));  EPI 1: returns this THREAD's share of
// sum (c*(x - P(u)))^2 over the item (chunk: elements t, t+256, ...; run of small blocks: thread t = block t's chain)
template <int EPI>
__device__ __forceinline__ double segs_apply_item(const ccqp_seg* __restrict__ table, const ccqp_segwork& w,
                                                  const double* __restrict__ part, double a, const double* x, double b,
                                                  const double* y, const double* z, double* out, double c,
                                                  double* lds4, double* bcast) {
  double racc = 0.0;
  if (w.type == 0) {
    if ((int)threadIdx.x < w.chunk) {
      const ccqp_seg sg = table[w.seg + threadIdx.x];
      if (sg.dim > 0 && sg.dim <= CCQP_SMALL_SEG_MAX) {
        const segs_in u = {x + sg.offset, y ? y + sg.offset : nullptr, a, b};
        if constexpr (EPI == 0) {
          emit_store e = {out + sg.offset, z ? z + sg.offset : nullptr};
          segs_small(sg, u, e);
        } else if constexpr (EPI == 1) {
          emit_residual e = {x + sg.offset, c, 0.0};
          segs_small(sg, u, e);
          racc = e.acc;
        } else {
          emit_notclose e = {0.0};
          segs_small(sg, u, e);
          racc = e.acc;
        }
      }
    }
    return racc;
  }
  const ccqp_seg sg = table[w.seg];
  int64_t chunk;
  int nblk;
  segs_geometry(sg.dim, &chunk, &nblk);
  const int64_t lo = (int64_t)w.chunk * chunk;
  int64_t hi = lo + chunk;
  if (hi > sg.dim) hi = sg.dim;
  const segs_in u = {x + sg.offset, y ? y + sg.offset : nullptr, a, b};
  const double* xs = x + sg.offset;
  const double* zs = z ? z + sg.offset : nullptr;
  double* os = out ? out + sg.offset : nullptr;
  if (sg.kind <= CCQP_OP_BOX) {
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double v = u(i);
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      const double p = ccqp_proj_elem(sg.kind, sg.flags, v, lb, ub);
      if constexpr (EPI == 0) {
        os[i] = zs ? p - zs[i] : p;
      } else if constexpr (EPI == 1) {
        const double d = xs[i] - p;
        const double t = c * d;
        racc = fma(t, t, racc);
      } else {
        racc += ccqp_isclose(v, p) ? 0.0 : 1.0;
      }
    }
  } else {
    const double n2 = segs_finish(part + w.pbase, nblk, lds4, bcast);
    segs_norm_rule rule;
    rule.setup(sg, n2, u(sg.dim - 1));
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double v = u(i);
      const double p = rule(v, i == sg.dim - 1);
      if constexpr (EPI == 0) {
        os[i] = zs ? p - zs[i] : p;
      } else if constexpr (EPI == 1) {
        const double d = xs[i] - p;
        const double t = c * d;
        racc = fma(t, t, racc);
      } else {
        racc += ccqp_isclose(v, p) ? 0.0 : 1.0;
      }
    }
  }
  return racc;
}

// One work item of an apply pass with a caller-supplied epilogue: emit(gi, u, p) receives, for every element of the
// item, its index in the whole vector, the input u = a*x + b*y and the projection P(u) -- computed exactly as
// segs_apply_item computes them (same block-norm finish, same Sphere / Cone rule), so anything built on this has the bits
// of the single-problem launches.  `emit` is per thread (it may accumulate).
template <class Emit>
__device__ __forceinline__ void segs_apply_emit(const ccqp_seg* __restrict__ table, const ccqp_segwork& w,
                                                const double* __restrict__ part, double a, const double* x, double b,
                                                const double* y, Emit& emit, double* lds4, double* bcast) {
  if (w.type == 0) {
    if ((int)threadIdx.x < w.chunk) {
      const ccqp_seg sg = table[w.seg + threadIdx.x];
      if (sg.dim > 0 && sg.dim <= CCQP_SMALL_SEG_MAX) {
        const segs_in u = {x + sg.offset, y ? y + sg.offset : nullptr, a, b};
        struct shifted {
          Emit& e;
          int64_t off;
          __device__ __forceinline__ void operator()(int64_t i, double uu, double pp) { e(off + i, uu, pp); }
        } sh = {emit, sg.offset};
        segs_small(sg, u, sh);
      }
    }
    return;
  }
  const ccqp_seg sg = table[w.seg];
  int64_t chunk;
  int nblk;
  segs_geometry(sg.dim, &chunk, &nblk);
  const int64_t lo = (int64_t)w.chunk * chunk;
  int64_t hi = lo + chunk;
  if (hi > sg.dim) hi = sg.dim;
  const segs_in u = {x + sg.offset, y ? y + sg.offset : nullptr, a, b};
  if (sg.kind <= CCQP_OP_BOX) {
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double v = u(i);
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      emit(sg.offset + i, v, ccqp_proj_elem(sg.kind, sg.flags, v, lb, ub));
    }
  } else {
    const double n2 = segs_finish(part + w.pbase, nblk, lds4, bcast);
    segs_norm_rule rule;
    rule.setup(sg, n2, u(sg.dim - 1));
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double v = u(i);
      emit(sg.offset + i, v, rule(v, i == sg.dim - 1));
    }
  }
}

// normal_vector of one work item, pass 2 of 3 (after the chunk sums of |x|^2): xp = P(x) into `out`, chunk sums of
// |x - xp|^2 (and |xp|^2 for Spheres) into dpart / lpart; small blocks finish here with their normal vector in `out`
// (solution_spaces.py:146-160,222-236,306-322,389-403,92-98).  extra(gi, x_i, xp_i) sees every element's projection
// before `out` is turned into the normal vector (the lock-step MPRGP derives psi and 1 - delta from it).
template <class Extra>
__device__ __forceinline__ void segs_normal_project_item(const ccqp_seg* __restrict__ table, const ccqp_segwork& w,
                                                         const double* __restrict__ part, double* __restrict__ dpart,
                                                         double* __restrict__ lpart, const double* x, double* out,
                                                         Extra& extra, double* lds4, double* bcast) {
  if (w.type == 0) {
    if ((int)threadIdx.x >= w.chunk) return;
    const ccqp_seg sg = table[w.seg + threadIdx.x];
    const int64_t d = sg.dim;
    if (d <= 0 || d > CCQP_SMALL_SEG_MAX) return;
    const double* xs = x + sg.offset;
    double* os = out + sg.offset;
    if (sg.kind == CCQP_OP_IDENTITY) {
      for (int64_t i = 0; i < d; ++i) {
        extra(sg.offset + i, xs[i], xs[i]);
        os[i] = 0.0;
      }
      return;
    }
    const segs_in u = {xs, nullptr, 1.0, 0.0};
    emit_store e = {os, nullptr};
    segs_small(sg, u, e);
    double dist2 = 0.0;
    for (int64_t i = 0; i < d; ++i) {
      extra(sg.offset + i, xs[i], os[i]);
      const double df = xs[i] - os[i];
      dist2 = fma(df, df, dist2);
    }
    const bool outside = !ccqp_isclose(sqrt(dist2), 0.0);
    if (sg.kind == CCQP_OP_SPHERE) {
      double l2 = 0.0;
      for (int64_t i = 0; i < d; ++i) l2 = fma(os[i], os[i], l2);
      const double len = sqrt(l2);
      const bool on = !outside && ccqp_isclose(len, sg.lb_s);
      for (int64_t i = 0; i < d; ++i) os[i] = on ? os[i] / len : 0.0;
      return;
    }
    const bool has_ub = sg.kind == CCQP_OP_UPPER || sg.kind == CCQP_OP_BOX;
    const bool has_lb = sg.kind == CCQP_OP_LOWER || sg.kind == CCQP_OP_BOX;
    for (int64_t i = 0; i < d; ++i) {
      double nv = 0.0;
      if (!outside) {
        const double xp = os[i];
        const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
        const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
        if (has_ub && ccqp_isclose(xp, ub)) nv = 1.0;
        else if (has_lb && ccqp_isclose(xp, lb)) nv = -1.0;
      }
      os[i] = nv;
    }
    return;
  }
  const ccqp_seg sg = table[w.seg];
  int64_t chunk;
  int nblk;
  segs_geometry(sg.dim, &chunk, &nblk);
  const int64_t lo = (int64_t)w.chunk * chunk;
  int64_t hi = lo + chunk;
  if (hi > sg.dim) hi = sg.dim;
  const double* xs = x + sg.offset;
  double* os = out + sg.offset;
  double dacc = 0.0, lacc = 0.0;
  if (sg.kind <= CCQP_OP_BOX) {
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double v = xs[i];
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      const double p = ccqp_proj_elem(sg.kind, sg.flags, v, lb, ub);
      extra(sg.offset + i, v, p);
      os[i] = p;
      const double df = v - p;
      dacc = fma(df, df, dacc);
    }
  } else {
    const double n2 = segs_finish(part + w.pbase, nblk, lds4, bcast);
    segs_norm_rule rule;
    rule.setup(sg, n2, xs[sg.dim - 1]);
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double v = xs[i];
      const double p = rule(v, i == sg.dim - 1);
      extra(sg.offset + i, v, p);
      os[i] = p;
      const double df = v - p;
      dacc = fma(df, df, dacc);
      lacc = fma(p, p, lacc);
    }
  }
  const double ds = ccqp_block_sum(dacc, lds4);
  const double ls = ccqp_block_sum(lacc, lds4);
  if (threadIdx.x == 0) {
    dpart[w.pbase + w.chunk] = ds;
    lpart[w.pbase + w.chunk] = ls;
  }
}

struct segs_no_extra {
  __device__ __forceinline__ void operator()(int64_t, double, double) const {}
};

// normal_vector, pass 3 of 3: large blocks only, in place on `out` (which holds xp)
__device__ __forceinline__ void segs_normal_finish_item(const ccqp_seg* __restrict__ table, const ccqp_segwork& w,
                                                        const double* __restrict__ dpart,
                                                        const double* __restrict__ lpart, double* out, double* lds4,
                                                        double* bcast) {
  if (w.type == 0) return;
  const ccqp_seg sg = table[w.seg];
  int64_t chunk;
  int nblk;
  segs_geometry(sg.dim, &chunk, &nblk);
  const int64_t lo = (int64_t)w.chunk * chunk;
  int64_t hi = lo + chunk;
  if (hi > sg.dim) hi = sg.dim;
  double* os = out + sg.offset;
  if (sg.kind == CCQP_OP_IDENTITY) {
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) os[i] = 0.0;
    return;
  }
  const double dist2 = segs_finish(dpart + w.pbase, nblk, lds4, bcast);
  const bool outside = !ccqp_isclose(sqrt(dist2), 0.0);
  if (sg.kind == CCQP_OP_SPHERE) {
    const double len = sqrt(segs_finish(lpart + w.pbase, nblk, lds4, bcast));
    const bool on = !outside && ccqp_isclose(len, sg.lb_s);
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) os[i] = on ? os[i] / len : 0.0;
    return;
  }
  const bool has_ub = sg.kind == CCQP_OP_UPPER || sg.kind == CCQP_OP_BOX;
  const bool has_lb = sg.kind == CCQP_OP_LOWER || sg.kind == CCQP_OP_BOX;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    double nv = 0.0;
    if (!outside) {
      const double v = os[i];
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      if (has_ub && ccqp_isclose(v, ub)) nv = 1.0;
      else if (has_lb && ccqp_isclose(v, lb)) nv = -1.0;
    }
    os[i] = nv;
  }
}

#endif  // __HIPCC__
