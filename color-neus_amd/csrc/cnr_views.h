// Operand views and epilogues of the two GEMM kernels (layer GEMM, weight-gradient GEMM).
//
// A "view" is a [P x K] fp32 operand = a plain row-major matrix in HBM (row stride a multiple of 4 floats, padded with
// finite values up to a multiple of 16 columns) plus a fused element-wise prologue (softplus, sigma'(z)*v, scaling ...)
// that is applied in registers while the tile travels HBM -> LDS.  The loads are unconditional 16-byte loads (no
// branches around memory operations in the GEMM main loop); only the math is interpreted (wave-uniform switch).
// An "epilogue" consumes 4 consecutive accumulator columns of one row and applies bias / activation / second-order
// terms / split stores with 16-byte accesses where the layout allows.
//
// Everything here is host+device so that the CPU emulation build (tests only) executes the same code.
#pragma once
#include "cnr_common.h"

namespace cnr {

enum ViewKind : int {
  VK_DIRECT = 0,   // a[row][col]
  VK_SOFTPLUS,     // softplus100(a[row][col])   for col < math_split, a[row][col] beyond (skip concat tail)
  VK_SIGMUL,       // softplus100'(a[row][col]) * b[row][col]                u_l = sp'(z_l) * v_l
  VK_SIGMUL_ROW,   // softplus100'(a[row][col]) * b[col]                     v_l is one broadcast row
  VK_CONST_COL0,   // col == hot ? 1 : 0 (hot = math_split, default 0)       u_top = unit vector of the sdf row (a must still be a valid pointer)
  VK_RELUGATE,     // b[row][col] > 0 ? a[row][col] : 0                      cotangent of a ReLU output gated by that output (cnr_linear_backward)
};

struct View {
  int kind = VK_DIRECT;
  const float* a = nullptr; int lda = 0;
  const float* b = nullptr; int ldb = 0;
  int math_split = 1 << 30;            // columns >= math_split bypass the kind's math (identity)
  float scale = 1.0f;                  // multiplies everything
};

struct Raw4 { f4 a; f4 b; };

// pure loads: 4 consecutive columns (col % 4 == 0, lda % 4 == 0) of one row
CNR_HD Raw4 view_fetch4(const View& v, long row, int col) {
  Raw4 r;
  r.a = *reinterpret_cast<const f4*>(v.a + row * v.lda + col);
  if (v.kind == VK_SIGMUL || v.kind == VK_RELUGATE) r.b = *reinterpret_cast<const f4*>(v.b + row * v.ldb + col);
  else if (v.kind == VK_SIGMUL_ROW) r.b = *reinterpret_cast<const f4*>(v.b + col);
  else { r.b.x = 0.f; r.b.y = 0.f; r.b.z = 0.f; r.b.w = 0.f; }   // (not a copy of r.a: that would make the caller wait for the load right here)
  return r;
}

// exact two-way select without a branch (the compiler otherwise wraps every element of a prologue in an exec-mask branch)
CNR_HD float select_f32(bool c, float x, float y) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CNR_CPU_EMU)
  const unsigned m = c ? 0xffffffffu : 0u;
  return __uint_as_float((__float_as_uint(x) & m) | (__float_as_uint(y) & ~m));
#else
  return c ? x : y;
#endif
}

CNR_HD float view_math1(const View& v, float a, float b, int col) {
  switch (v.kind) {
    case VK_DIRECT: return a;
    case VK_SOFTPLUS: return select_f32(col < v.math_split, softplus100(a), a);
    case VK_SIGMUL:
    case VK_SIGMUL_ROW: return softplus100_d1(a) * b;
    case VK_RELUGATE: return b > 0.0f ? a : 0.0f;
    default: return col == (v.math_split == (1 << 30) ? 0 : v.math_split) ? 1.0f : 0.0f;
  }
}

// pure math
CNR_HD f4 view_finish4(const View& v, const Raw4& raw, int col) {
  f4 r;
  r.x = view_math1(v, raw.a.x, raw.b.x, col);
  r.y = view_math1(v, raw.a.y, raw.b.y, col + 1);
  r.z = view_math1(v, raw.a.z, raw.b.z, col + 2);
  r.w = view_math1(v, raw.a.w, raw.b.w, col + 3);
  if (v.scale != 1.0f) { r.x *= v.scale; r.y *= v.scale; r.z *= v.scale; r.w *= v.scale; }
  return r;
}

CNR_HD f4 view_eval4(const View& v, long row, int col) { return view_finish4(v, view_fetch4(v, row, col), col); }

// ------------------------------------------------------------------------------------------------
enum EpiKind : int {
  EK_STORE = 0,    // o1[row][o1_off+col] = (acc + bias[col]) * scale ; optional tail fill (see tail_*)
  EK_SPLIT,        // v = (acc+bias)*scale ; col < split -> o1[row][o1_off+col] ; else o2[row][col-split]   (o2 may be null)
  EK_SDF_TOP,      // internal row order [features | sdf]: col < split -> o1[row][col] = acc+bias ; col == split -> o2[row] = (acc+bias)*scale
  EK_RELU,         // o1 = relu(acc + bias)
  EK_SIGMOID,      // y = sigmoid(acc + bias) -> o1[row][col] ; optional copy o2[row][o2_off+col] (zero for col >= n_out, col < 16)
  EK_LINEAR_SIG,   // as EK_SIGMOID without the sigmoid (unsqueezed colour output)
  EK_RELIGHT_TOP,  // t = acc+bias ; o1[row][col] = t ; o2[row][col] = relight(aux[row][col], t)
  EK_SWEEP,        // u = acc ; o1 = sp''(z) * v * u  (second-order cotangent on z) ; o2 = sp'(z) * u (tangent of h) ; tail fill on o2
  EK_VBACK,        // h = acc*scale ; col < split -> o1[row][col] = sp'(z[row][col]) * h + o1[row][col] ; else o2[row][col-split] = h
  EK_RELU_MASK,    // v = acc ; col < split -> o1[row][col] = aux[row][col] > 0 ? v : 0 ; else o2[row][col-split] = v
};

struct Epi {
  int kind = EK_STORE;
  int n_out = 0;                       // logical number of output columns (cols >= n_out are dropped)
  const float* bias = nullptr;
  float scale = 1.0f;
  float* o1 = nullptr; int ld1 = 0; int o1_off = 0;
  float* o2 = nullptr; int ld2 = 0; int o2_off = 0;
  bool o2_acc = false;   // EK_SPLIT / EK_VBACK: ADD the beyond-split columns to o2 instead of storing them (the embedding cotangent of the 2nd, 3rd .. skip connection)
  int split = 1 << 30;
  const float* z = nullptr; int ldz = 0;      // pre-activations (EK_SWEEP / EK_VBACK)
  const float* v = nullptr; int ldv = 0;      // grad-chain cotangent v_l (EK_SWEEP); ldv == 0 -> broadcast row
  float vscale = 1.0f;                        // multiplies v (the broadcast row is W_top[0,:] / scale)
  const float* aux = nullptr; int ldaux = 0;  // relu mask source / global colour
  int inv_sigmoid = 1;                        // EK_RELIGHT_TOP mode
  // tail fill: columns [n_out, n_out + tail_n) of the main output row (o1 for EK_STORE, o2 for EK_SWEEP) receive
  // tail_src[row][col - n_out] -- this is how the skip-connection concat [h | e] is materialised without a copy kernel
  const float* tail_src = nullptr; int ld_tail = 0; int tail_n = 0;
};

// rgb' = sigmoid(inverse_sigmoid(rgb) + t)   (reference fields.py:354-359, transform.py:304-320)
CNR_HD float relight_apply(float rgb, float t, int inv_sigmoid) {
  if (inv_sigmoid) {
    float x = fminf(fmaxf(rgb, 0.0f), 1.0f);
    float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.0f - x, 1e-5f);
    return sigmoidf_(logf(x1 / x2) + t);
  }
  return fminf(fmaxf(rgb + sigmoidf_(t) - 0.5f, 0.0f), 1.0f);
}

// one accumulator element
CNR_HD void epi_apply(const Epi& e, long row, int col, float acc) {
  if (col >= e.n_out) {
    if (e.tail_src && col < e.n_out + e.tail_n) {
      float t = e.tail_src[row * e.ld_tail + (col - e.n_out)];
      if (e.kind == EK_STORE) e.o1[row * e.ld1 + e.o1_off + col] = t;
      else if (e.kind == EK_SWEEP) { if (e.o2) e.o2[row * e.ld2 + col] = t; e.o1[row * e.ld1 + col] = 0.0f; }
    }
    if ((e.kind == EK_SIGMOID || e.kind == EK_LINEAR_SIG) && e.o2 && col < 16) e.o2[row * e.ld2 + e.o2_off + col] = 0.0f;
    return;
  }
  switch (e.kind) {
    case EK_STORE: {
      float b = e.bias ? e.bias[col] : 0.0f;
      e.o1[row * e.ld1 + e.o1_off + col] = (acc + b) * e.scale;
    } break;
    case EK_SPLIT: {
      float b = e.bias ? e.bias[col] : 0.0f;
      float v = (acc + b) * e.scale;
      if (col < e.split) e.o1[row * e.ld1 + e.o1_off + col] = v;
      else if (e.o2) { float* q = e.o2 + row * e.ld2 + (col - e.split) + e.o2_off; *q = e.o2_acc ? *q + v : v; }
    } break;
    case EK_SDF_TOP: {
      float v = acc + e.bias[col];
      if (col < e.split) { if (e.o1) e.o1[row * e.ld1 + col] = v; }
      else e.o2[row] = v * e.scale;
    } break;
    case EK_RELU: {
      float v = acc + e.bias[col];
      e.o1[row * e.ld1 + col] = v > 0.0f ? v : 0.0f;
    } break;
    case EK_SIGMOID:
    case EK_LINEAR_SIG: {
      float y = acc + e.bias[col];
      if (e.kind == EK_SIGMOID) y = sigmoidf_(y);
      e.o1[row * e.ld1 + col] = y;
      if (e.o2) e.o2[row * e.ld2 + e.o2_off + col] = y;
    } break;
    case EK_RELIGHT_TOP: {
      float t = acc + e.bias[col];
      e.o1[row * e.ld1 + col] = t;
      e.o2[row * e.ld2 + col] = relight_apply(e.aux[row * e.ldaux + col], t, e.inv_sigmoid);
    } break;
    case EK_SWEEP: {
      float zz = e.z[row * e.ldz + col];
      float vv = (e.ldv ? e.v[row * e.ldv + col] : e.v[col]) * e.vscale;
      e.o1[row * e.ld1 + col] = softplus100_d2(zz) * vv * acc;
      if (e.o2) e.o2[row * e.ld2 + col] = softplus100_d1(zz) * acc;
    } break;
    case EK_VBACK: {
      float h = acc * e.scale;
      if (col < e.split) {
        float* p = e.o1 + row * e.ld1 + col;
        *p = softplus100_d1(e.z[row * e.ldz + col]) * h + *p;
      } else if (e.o2) {
        float* q = e.o2 + row * e.ld2 + (col - e.split) + e.o2_off;
        *q = e.o2_acc ? *q + h : h;
      }
    } break;
    default: {  // EK_RELU_MASK
      if (col < e.split) e.o1[row * e.ld1 + col] = e.aux[row * e.ldaux + col] > 0.0f ? acc : 0.0f;
      else if (e.o2) e.o2[row * e.ld2 + (col - e.split)] = acc;
    } break;
  }
}

// 4 consecutive accumulator columns (col % 4 == 0): 16-byte fast paths for the bandwidth-heavy kinds, element-wise otherwise
// ---- 16-byte epilogue path, split into "is it applicable" / "side loads" / "math + stores" so that a kernel can issue the
// side loads of several rows back to back (one memory round trip per tile instead of one per row group)
struct EpiRaw4 { f4 a; f4 b; };   // (launches with a tail source: the elements of a at columns >= n_out hold the tail-fill values)

// true when 4 consecutive columns starting at col (col % 4 == 0) take the 16-byte path; depends on the column only
CNR_HD bool epi_fast4(const Epi& e, int col) {
  // groups that straddle n_out or lie in the tail-fill range stay on this path too (EK_STORE / EK_SWEEP with a tail source)
  const bool tail_ok = e.tail_src != nullptr && (e.kind == EK_STORE || e.kind == EK_SWEEP) && col + 4 <= e.n_out + e.tail_n;
  // EK_SPLIT / EK_VBACK groups at or beyond the split point: the o2 columns land 16-byte aligned when o2_off == split % 4
  // (the consumer reads o2 + o2_off); a straddling EK_SPLIT group also writes its beyond-split values into the pad columns of o1
  const bool split_ok = (e.kind == EK_SPLIT || e.kind == EK_VBACK) && ((e.o2_off - e.split) & 3) == 0 && (e.ld2 & 3) == 0 &&
                        (e.kind == EK_VBACK || e.o1_off == 0);
  if (!((col + 4 <= e.n_out || tail_ok) && (col + 4 <= e.split || split_ok))) return false;
  switch (e.kind) {
    case EK_SDF_TOP: return (e.ld1 & 3) == 0 && e.o1 != nullptr;
    case EK_SPLIT:
    case EK_STORE: return ((e.ld1 | e.o1_off) & 3) == 0;
    case EK_RELU: return (e.ld1 & 3) == 0;
    case EK_SWEEP: return ((e.ld1 | e.ld2 | e.ldz | e.ldv) & 3) == 0 && e.o2 != nullptr;
    case EK_VBACK: return ((e.ld1 | e.ldz) & 3) == 0;
    case EK_RELU_MASK: return ((e.ld1 | e.ldaux) & 3) == 0;
    default: return false;
  }
}
// the per-row side inputs of the fast path (only valid when epi_fast4)
CNR_HD EpiRaw4 epi_fetch4(const Epi& e, long row, int col) {
  EpiRaw4 r;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  r.a = zero; r.b = zero;
  switch (e.kind) {
    case EK_SWEEP:
      if (col < e.n_out) {
        r.a = *reinterpret_cast<const f4*>(e.z + row * e.ldz + col);
        r.b = e.ldv ? *reinterpret_cast<const f4*>(e.v + row * e.ldv + col) : *reinterpret_cast<const f4*>(e.v + col);
      }
      break;
    case EK_VBACK:
      if (col < e.split) {
        r.a = *reinterpret_cast<const f4*>(e.z + row * e.ldz + col);
        r.b = *reinterpret_cast<const f4*>(e.o1 + row * e.ld1 + col);
      }
      break;
    case EK_RELU_MASK:
      r.a = *reinterpret_cast<const f4*>(e.aux + row * e.ldaux + col);
      break;
    default: break;
  }
  if (e.tail_src && col + 4 > e.n_out && (e.kind == EK_STORE || e.kind == EK_SWEEP)) {   // fetched with the other side inputs, ahead of use
    const float* t = e.tail_src + row * e.ld_tail - e.n_out;
    if (col >= e.n_out) r.a.x = t[col];
    if (col + 1 >= e.n_out) r.a.y = t[col + 1];
    if (col + 2 >= e.n_out) r.a.z = t[col + 2];
    if (col + 3 >= e.n_out) r.a.w = t[col + 3];
  }
  return r;
}
// bias of 4 columns (zero when the epilogue has none)
CNR_HD f4 epi_bias4(const Epi& e, int col) {
  f4 b = {0.f, 0.f, 0.f, 0.f};
  if (!e.bias) return b;
  if (col + 4 <= e.n_out) return *reinterpret_cast<const f4*>(e.bias + col);
  if (col < e.n_out) b.x = e.bias[col];
  if (col + 1 < e.n_out) b.y = e.bias[col + 1];
  if (col + 2 < e.n_out) b.z = e.bias[col + 2];
  if (col + 3 < e.n_out) b.w = e.bias[col + 3];
  return b;
}
CNR_HD void epi_finish4(const Epi& e, long row, int col, const f4& acc, const f4& b, const EpiRaw4& raw) {
  switch (e.kind) {
    case EK_SDF_TOP: {
      f4 o;
      o.x = acc.x + b.x; o.y = acc.y + b.y; o.z = acc.z + b.z; o.w = acc.w + b.w;
      *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = o;
    } break;
    case EK_SPLIT: {
      f4 o;
      o.x = (acc.x + b.x) * e.scale; o.y = (acc.y + b.y) * e.scale; o.z = (acc.z + b.z) * e.scale; o.w = (acc.w + b.w) * e.scale;
      if (col < e.split) *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + e.o1_off + col) = o;
      if (col + 4 > e.split && e.o2) {
        f4* q = reinterpret_cast<f4*>(e.o2 + row * e.ld2 + (col - e.split) + e.o2_off);
        if (e.o2_acc) { const f4 old = *q; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }   // (a straddling group also touches the pad columns in front of o2_off: never read)
        *q = o;
      }
    } break;
    case EK_STORE: {
      f4 o;
      o.x = (acc.x + b.x) * e.scale; o.y = (acc.y + b.y) * e.scale; o.z = (acc.z + b.z) * e.scale; o.w = (acc.w + b.w) * e.scale;
      if (e.tail_src && col + 4 > e.n_out) {   // tail-fill columns (only the launches that materialise a skip concat have them)
        if (col >= e.n_out) o.x = raw.a.x;
        if (col + 1 >= e.n_out) o.y = raw.a.y;
        if (col + 2 >= e.n_out) o.z = raw.a.z;
        if (col + 3 >= e.n_out) o.w = raw.a.w;
      }
      *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + e.o1_off + col) = o;
    } break;
    case EK_RELU: {
      f4 o;
      o.x = fmaxf(acc.x + b.x, 0.f); o.y = fmaxf(acc.y + b.y, 0.f); o.z = fmaxf(acc.z + b.z, 0.f); o.w = fmaxf(acc.w + b.w, 0.f);
      *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = o;
    } break;
    case EK_SWEEP: {
      const f4 zz = raw.a, vv = raw.b;
      f4 o1, o2;
      o1.x = softplus100_d2(zz.x) * (vv.x * e.vscale) * acc.x; o2.x = softplus100_d1(zz.x) * acc.x;
      o1.y = softplus100_d2(zz.y) * (vv.y * e.vscale) * acc.y; o2.y = softplus100_d1(zz.y) * acc.y;
      o1.z = softplus100_d2(zz.z) * (vv.z * e.vscale) * acc.z; o2.z = softplus100_d1(zz.z) * acc.z;
      o1.w = softplus100_d2(zz.w) * (vv.w * e.vscale) * acc.w; o2.w = softplus100_d1(zz.w) * acc.w;
      if (e.tail_src && col + 4 > e.n_out) {
        if (col >= e.n_out) { o1.x = 0.f; o2.x = raw.a.x; }
        if (col + 1 >= e.n_out) { o1.y = 0.f; o2.y = raw.a.y; }
        if (col + 2 >= e.n_out) { o1.z = 0.f; o2.z = raw.a.z; }
        if (col + 3 >= e.n_out) { o1.w = 0.f; o2.w = raw.a.w; }
      }
      *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = o1;
      *reinterpret_cast<f4*>(e.o2 + row * e.ld2 + col) = o2;
    } break;
    case EK_VBACK: {
      f4 h;
      h.x = acc.x * e.scale; h.y = acc.y * e.scale; h.z = acc.z * e.scale; h.w = acc.w * e.scale;
      if (col < e.split) {
        const f4 zz = raw.a;
        f4 o = raw.b;
        o.x = softplus100_d1(zz.x) * h.x + o.x;
        if (col + 1 < e.split) o.y = softplus100_d1(zz.y) * h.y + o.y;
        if (col + 2 < e.split) o.z = softplus100_d1(zz.z) * h.z + o.z;
        if (col + 3 < e.split) o.w = softplus100_d1(zz.w) * h.w + o.w;
        *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = o;
      }
      if (col + 4 > e.split && e.o2) {
        f4* q = reinterpret_cast<f4*>(e.o2 + row * e.ld2 + (col - e.split) + e.o2_off);
        if (e.o2_acc) { const f4 old = *q; h.x += old.x; h.y += old.y; h.z += old.z; h.w += old.w; }
        *q = h;
      }
    } break;
    case EK_RELU_MASK: {
      const f4 m = raw.a;
      f4 o;
      o.x = m.x > 0.f ? acc.x : 0.f; o.y = m.y > 0.f ? acc.y : 0.f; o.z = m.z > 0.f ? acc.z : 0.f; o.w = m.w > 0.f ? acc.w : 0.f;
      *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = o;
    } break;
    default: break;
  }
}

CNR_HD void epi_apply4(const Epi& e, long row, int col, f4 acc) {
  if (epi_fast4(e, col)) {
    epi_finish4(e, row, col, acc, epi_bias4(e, col), epi_fetch4(e, row, col));
    return;
  }
  if (col >= e.n_out && e.tail_src && col + 4 <= e.n_out + e.tail_n) {   // tail fill: 4 scalar reads (source is unaligned), one 16-byte store
    const float* t = e.tail_src + row * e.ld_tail + (col - e.n_out);
    const f4 tv = {t[0], t[1], t[2], t[3]};
    if (e.kind == EK_STORE && ((e.ld1 | e.o1_off) & 3) == 0) {
      *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + e.o1_off + col) = tv;
      return;
    }
    if (e.kind == EK_SWEEP && ((e.ld1 | e.ld2) & 3) == 0 && e.o2) {
      const f4 zero = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f4*>(e.o2 + row * e.ld2 + col) = tv;
      *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = zero;
      return;
    }
  }
  epi_apply(e, row, col, acc.x);
  epi_apply(e, row, col + 1, acc.y);
  epi_apply(e, row, col + 2, acc.z);
  epi_apply(e, row, col + 3, acc.w);
}

// ------------------------------------------------------------------------------------------------
// C[P x N] = epilogue( A[P x K] * W[N x K]^T )
struct LayerGemm {
  View A;
  const float* W = nullptr;   // [round_up(N,32)][ldw], zero padded, ldw = round_up(K,16)
  int ldw = 0;
  int N = 0, K = 0;
  long P = 0;
  Epi E;
  int col0 = 0;               // first output column handled by this launch (a wide layer may be split into column ranges)
  int first_col = 0;          // be_layer_gemm starts at this column (a multiple of 256): the columns below belong to another launch
  const unsigned short* Wp = nullptr;   // optional: W as two f16 planes (hi, lo) of the row-scaled weights, same [rows][ldw] layout each
  long wp_stride = 0;                   // elements between planes
  int w_rows = 0;                       // rows of W / Wp that exist (zero rows beyond N); 0: round_up(N, 32)
  int k_extra = 0;                      // 1: the contraction has one more index K (A column K, W column K): a launch may take it as a rank-one update
                                        // of the product in its epilogue (fused layer + weight-gradient launch of the 257-wide top SDF layer); others add it to K
  const float* wscale = nullptr;        // per W row: 1 / (power-of-two scale applied before the f16 split)
  const int* P_dev = nullptr; // optional device-side row count (<= P): compacted point lists whose length only the GPU knows
  // optional row dot product formed while the input tile is staged (the 16 threads that stage a row hold all of it): the ONE extra output
  // column of a 257-wide layer (the sdf row of the top SDF layer) without a second launch over the same rows:
  //   dot_out[row] = (sum_k A[row][k] * dot_w[k] + dot_bias[0]) * dot_scale        (fp32 FMAs, fixed order)
  const float* dot_w = nullptr; const float* dot_bias = nullptr; float dot_scale = 1.0f; float* dot_out = nullptr;
  float* rs_out = nullptr;    // optional [P]: the power-of-two scale that lifts each (prologue-applied) operand row into the top f16
                              // binade, 0 for an all-zero row; the weight-gradient GEMM re-uses it for the same operand
};

// dW[N x K] (+)= sum over points of X[pt][n] * Y[pt][k]   (up to two operand pairs share the accumulators)
struct DwGemm {
  View X[2];
  View Y[2];
  int npairs = 1;
  int N = 0, K = 0;
  long P = 0;
  int nchunk = 1;             // the point range is cut into nchunk slices, one partial result each
  long chunk_pts = 0;         // multiple of 16
  float* partial = nullptr;   // [nchunk][Npad][ldk]
  int Npad = 0, ldk = 0;
  float* colsum = nullptr;    // optional [nchunk][Npad]: column sums of X[0] (bias gradient)
  // optional per-point row scales of the operands (LayerGemm::rs_out of the launch that consumed the same view).  With all of them
  // present (split_f16) the 256 x 256 tiles run as f16 x 3; every workgroup takes the exponent G = 1 + min log2(sx * sy) over its own points
  const float* sx[2] = {nullptr, nullptr};
  const float* sy[2] = {nullptr, nullptr};
  bool split_f16 = false;
  bool skip_main = false;     // leave out the 256 x 256 main tiles (formed elsewhere: the fused layer + weight-gradient launch); strips only
};

// Algorithmic HBM bytes of one launch: every operand matrix read once, every output written once (weights and bias are
// noise at these sizes and are left out).  Used only for the roofline figures of the timing records.
inline double view_bytes(const View& v, long P, int K) {
  switch (v.kind) {
    case VK_SIGMUL: case VK_RELUGATE: return 8.0 * (double)P * K;
    case VK_CONST_COL0: return 0.0;
    default: return 4.0 * (double)P * K;
  }
}
inline double layer_gemm_bytes(const LayerGemm& g) {
  const Epi& e = g.E;
  const double P = (double)g.P;
  const int lo = g.col0, hi = g.col0 + g.N;                     // output column range of this launch
  auto span = [&](int a, int b) { int x = a > lo ? a : lo, y = b < hi ? b : hi; return y > x ? (double)(y - x) : 0.0; };
  const double n = span(0, e.n_out), nlo = span(0, e.split < e.n_out ? e.split : e.n_out), nhi = n - nlo;
  const double tail = e.tail_src ? span(e.n_out, e.n_out + e.tail_n) : 0.0;
  double b = view_bytes(g.A, g.P, g.K);
  switch (e.kind) {
    case EK_STORE: b += 4 * P * (n + 2 * tail); break;
    case EK_SPLIT: b += 4 * P * (nlo + (e.o2 ? nhi : 0.0)); break;
    case EK_SDF_TOP: b += 4 * P * ((e.o1 ? nlo : 0.0) + nhi); break;
    case EK_RELU: b += 4 * P * n; break;
    case EK_SIGMOID: case EK_LINEAR_SIG: b += 4 * P * n * (e.o2 ? 2 : 1); break;
    case EK_RELIGHT_TOP: b += 4 * P * n * 3; break;
    case EK_SWEEP: b += 4 * P * (n * (e.ldv ? 4 : 3) + 2 * tail); break;
    case EK_VBACK: b += 4 * P * (3 * nlo + nhi); break;
    case EK_RELU_MASK: b += 4 * P * (2 * nlo + (e.o2 ? nhi : 0.0)); break;
  }
  return b;
}
// Layer GEMM + weight gradient in ONE launch (cnr_gemm_fdw.hip): the launch that consumes an operand S = A (its staged input) forms the
// weight-gradient pair (S, Ep) on chip, where Ep ("epilogue-side operand") is what its fused epilogue computes from its side inputs anyway:
//   EK_RELU_MASK  Ep = aux                         (the layer input h_{l-1} of the ReLU stacks; dW_l = sum_pt dout_l (x) h_{l-1})
//   EK_VBACK      Ep = softplus(z_{l-1})           (SDF value pair:           dW_l += sum_pt zbar_l (x) h_{l-1})
//   EK_SWEEP      Ep = sp'(z_l) * v_l * vscale     (SDF gradient-chain pair:  dW_l += sum_pt u_l (x) qbar_l, written transposed)
// so the separate weight-gradient GEMM of that pair -- a second pass over both operands -- disappears.
constexpr int kFdwSlots = 128;   // most point ranges (= partial-sum slots) of a fused launch: two workgroups (column halves) per range
struct DwFuse {
  const float* se = nullptr;  // [P] power-of-two row scales of Ep saved by the forward launch that consumed the same rows (LayerGemm::rs_out):
                              // 0 = all-zero row, NaN = non-finite row
  float* partial = nullptr;   // [nslots][Npad][ldk] partial sums of this pair, every slot written in full
  int Npad = 0, ldk = 0;
  float* colsum = nullptr;    // optional [nslots][Npad]: column sums of S (bias gradient); only with transposed == 0
  int transposed = 0;         // 0: dW[n = column of S][k = column of Ep] ; 1: dW[n = column of Ep][k = column of S]
  int nslots = 0;             // point ranges = partial-sum slots of this launch: a multiple of 8, <= kFdwSlots
  int rev = 0;                // 1: every range is walked downwards (see cnr_gemm_fdw.hip); the separate-kernel fallback ignores it
  // One extra weight-gradient row formed from values the epilogue holds anyway: the sdf row (internal row 256) of the 257-wide top SDF layer.
  //   xrow_mode 1 (EK_SWEEP launch of the layer below):   xrow[range][c]  = xrow_scale * sum over the range's points of o2[pt][c]
  //   xrow_mode 2 (EK_VBACK launch with k_extra == 1):    xrow[range][c] += sum of A[pt][K] * Ep[pt][c];  xbias[range] = sum of A[pt][K]
  // (mode 1 runs first and writes every slot row in full; mode 2 adds to it).  xrow + range * xrow_stride, 256 floats; xbias + range * xbias_stride.
  int xrow_mode = 0;
  float* xrow = nullptr; long xrow_stride = 0; float xrow_scale = 1.0f;
  float* xbias = nullptr; long xbias_stride = 0;
};

// structural conditions of the fused kernel: a plain 256 -> 256 launch on full 32-point tiles whose epilogue takes the 16-byte path everywhere
inline bool fdw_shape_ok(const LayerGemm& g) {
  const Epi& e = g.E;
  if (g.A.kind != VK_DIRECT || (g.A.lda & 3) != 0 || g.col0 != 0 || g.first_col != 0 || g.P_dev != nullptr) return false;
  if (g.P <= 0 || (g.P % 32) != 0 || g.Wp == nullptr || g.wscale == nullptr) return false;
  // (EK_SPLIT: colour layer 0 -- a plain scaled store; its epilogue has no side input, so the epilogue-side operand, the layer's forward
  // input, is named by aux / ldaux)
  if (e.kind != EK_RELU_MASK && e.kind != EK_VBACK && e.kind != EK_SWEEP && e.kind != EK_SPLIT) return false;
  // contraction: 16 k16 blocks, or 14 for the value-backward launch of the 217-wide layer (its pad columns are zero, its weight planes 224 wide)
  if (!(g.K > 240 && g.K <= 256) && !(e.kind == EK_VBACK && g.K > 208 && g.K <= 224 && g.ldw >= 224)) return false;
  // the first 256 output columns are plain (no tail fill, no split point below 256); further columns (N > 256, EK_RELU_MASK only: the
  // relight y-layer) go to the narrow launch that follows
  // ... or, for the sweep launch of the layer below a skip connection, n_out live columns + the tail fill up to column 256 (zero weight rows
  // under the tail columns: w_rows >= 256); that launch runs the general 16-byte epilogue code
  const bool tail_form = e.kind == EK_SWEEP && e.tail_src != nullptr && e.n_out + e.tail_n == 256 && e.n_out >= 192 && g.w_rows >= 256 &&
                         g.N == e.n_out && e.split == (1 << 30) && e.ldv != 0;
  // ... or, for the value-backward launch of the layer fed by a skip connection, a split point inside the 256 columns: the columns below it are
  // the hidden part (value-backward epilogue), those at and beyond it the embedding part (plain store to o2, whose 16-byte groups line up:
  // o2_off == split % 4); the epilogue-side operand is then [softplus(z) | z] * vscale, the layer's forward input as its view forms it
  const bool split_form = e.kind == EK_VBACK && e.tail_src == nullptr && e.n_out == 256 && g.N == 256 && e.split >= 192 && e.split < 256 &&
                          ((e.o2_off - e.split) & 3) == 0 && (e.ld2 & 3) == 0 && g.K > 240;
  if (!tail_form && !split_form && (e.tail_src != nullptr || e.n_out < 256 || e.split < 256 || g.N < 256)) return false;
  if (g.N > 256 && e.kind != EK_RELU_MASK) return false;
  switch (e.kind) {
    case EK_RELU_MASK: return ((e.ld1 | e.ldaux) & 3) == 0 && e.aux != nullptr;
    case EK_SPLIT: return ((e.ld1 | e.ldaux | e.o1_off) & 3) == 0 && e.aux != nullptr && e.o1 != nullptr && e.bias == nullptr;
    case EK_VBACK: return ((e.ld1 | e.ldz) & 3) == 0;
    default: return ((e.ld1 | e.ld2 | e.ldz | e.ldv) & 3) == 0 && e.o2 != nullptr;
  }
}
inline double fdw_bytes(const LayerGemm& g, const DwFuse& f) {   // the layer launch's operands + the partial sums; S counted once (its second read is an L2 hit)
  // (EK_SPLIT has no epilogue side input: its epilogue-side operand, aux, is an extra read of the fused launch)
  return layer_gemm_bytes(g) + (g.E.kind == EK_SPLIT ? 4.0 * (double)g.P * 256 : 0.0) + 4.0 * f.nslots * (double)f.Npad * f.ldk;
}

inline double dw_gemm_bytes(const DwGemm& g, int n, int k) {
  double b = 0;
  for (int i = 0; i < g.npairs; ++i) b += view_bytes(g.X[i], g.P, n) + view_bytes(g.Y[i], g.P, k);
  return b + 4.0 * g.nchunk * (double)n * k;
}

}  // namespace cnr
