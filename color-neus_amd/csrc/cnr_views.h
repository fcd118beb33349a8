// Operand views and epilogues of the two GEMM kernels (layer GEMM, weight-gradient GEMM).
//
// A "view" is a lazily evaluated [P x ncols] fp32 matrix: the GEMM staging code asks it for 4
// consecutive columns of one row and the view applies the fused prologue (softplus, sigma'(z)*v,
// concat of two sources, scaling ...) while the tile travels HBM -> LDS.  An "epilogue" consumes one
// accumulator element (row, col) and applies bias / activation / second-order terms / split stores.
// Both are interpreted (wave-uniform switch) so that ONE compiled kernel per tile shape serves every
// layer of the SDF / colour / relight stacks, forward and backward.
//
// Everything here is host+device so that the CPU emulation build (tests only) executes the same code.
#pragma once
#include "cnr_common.h"

namespace cnr {

enum ViewKind : int {
  VK_DIRECT = 0,   // a[row*lda + col]
  VK_SOFTPLUS,     // softplus100(a[row*lda + col])                         h_l = sp(z_l)
  VK_SIGMUL,       // softplus100'(a[row*lda+col]) * b[row*ldb + col]       u_l = sp'(z_l) * v_l
  VK_SIGMUL_ROW,   // softplus100'(a[row*lda+col]) * b[col]                 v_l is one broadcast row
  VK_CONST_COL0,   // col == 0 ? 1 : 0                                      u_top = e_0
};

struct View {
  int kind = VK_DIRECT;
  const float* a = nullptr; int lda = 0;
  const float* b = nullptr; int ldb = 0;
  int split = 1 << 30;                 // columns >= split come from the secondary (direct) source c
  const float* c = nullptr; int ldc = 0;
  int ncols = 0;                       // logical width; columns >= ncols evaluate to 0
  float scale = 1.0f;                  // multiplies primary and secondary
};

CNR_HD float view_primary1(const View& v, long row, int col) {
  switch (v.kind) {
    case VK_DIRECT: return v.a[row * v.lda + col];
    case VK_SOFTPLUS: return softplus100(v.a[row * v.lda + col]);
    case VK_SIGMUL: return softplus100_d1(v.a[row * v.lda + col]) * v.b[row * v.ldb + col];
    case VK_SIGMUL_ROW: return softplus100_d1(v.a[row * v.lda + col]) * v.b[col];
    default: return col == 0 ? 1.0f : 0.0f;
  }
}

CNR_HD float view_eval1(const View& v, long row, int col) {
  if (col >= v.ncols) return 0.0f;
  float x = col < v.split ? view_primary1(v, row, col) : v.c[row * v.ldc + (col - v.split)];
  return x * v.scale;
}

// 4 consecutive columns (col % 4 == 0).  row must be < nrows (caller guards).
CNR_HD f4 view_eval4(const View& v, long row, int col) {
  f4 r;
  int lim = v.split < v.ncols ? v.split : v.ncols;
  if (col + 4 <= lim && (v.lda & 3) == 0 && v.kind != VK_CONST_COL0) {
    const f4 za = *reinterpret_cast<const f4*>(v.a + row * v.lda + col);
    switch (v.kind) {
      case VK_DIRECT: r = za; break;
      case VK_SOFTPLUS:
        r.x = softplus100(za.x); r.y = softplus100(za.y); r.z = softplus100(za.z); r.w = softplus100(za.w);
        break;
      case VK_SIGMUL: {
        f4 vb;
        if ((v.ldb & 3) == 0) vb = *reinterpret_cast<const f4*>(v.b + row * v.ldb + col);
        else { vb.x = v.b[row * v.ldb + col]; vb.y = v.b[row * v.ldb + col + 1]; vb.z = v.b[row * v.ldb + col + 2]; vb.w = v.b[row * v.ldb + col + 3]; }
        r.x = softplus100_d1(za.x) * vb.x; r.y = softplus100_d1(za.y) * vb.y;
        r.z = softplus100_d1(za.z) * vb.z; r.w = softplus100_d1(za.w) * vb.w;
      } break;
      default: {  // VK_SIGMUL_ROW
        r.x = softplus100_d1(za.x) * v.b[col]; r.y = softplus100_d1(za.y) * v.b[col + 1];
        r.z = softplus100_d1(za.z) * v.b[col + 2]; r.w = softplus100_d1(za.w) * v.b[col + 3];
      } break;
    }
    if (v.scale != 1.0f) { r.x *= v.scale; r.y *= v.scale; r.z *= v.scale; r.w *= v.scale; }
    return r;
  }
  if (col >= v.ncols) { r.x = r.y = r.z = r.w = 0.0f; return r; }
  if (col >= v.split && col + 4 <= v.ncols && ((v.ldc | v.split) & 3) == 0) {
    r = *reinterpret_cast<const f4*>(v.c + row * v.ldc + (col - v.split));
    if (v.scale != 1.0f) { r.x *= v.scale; r.y *= v.scale; r.z *= v.scale; r.w *= v.scale; }
    return r;
  }
  r.x = view_eval1(v, row, col); r.y = view_eval1(v, row, col + 1);
  r.z = view_eval1(v, row, col + 2); r.w = view_eval1(v, row, col + 3);
  return r;
}

// ------------------------------------------------------------------------------------------------
enum EpiKind : int {
  EK_STORE = 0,    // o1[row][o1_off+col] = (acc + bias[col]) * scale
  EK_SPLIT,        // v = (acc+bias)*scale ; col < split -> o1[row][o1_off+col] ; else o2[row][col-split]   (o2 may be null)
  EK_SDF_TOP,      // col == 0 -> o2[row] = (acc+bias)*scale ; col >= 1 -> o1[row][col-1] = acc+bias
  EK_RELU,         // o1 = relu(acc + bias)
  EK_SIGMOID,      // o1 = sigmoid(acc + bias)
  EK_LINEAR_SIG,   // o1 = acc + bias (unsqueezed colour output)
  EK_RELIGHT_TOP,  // t = acc+bias ; o1[row][col] = t ; o2[row][col] = relight(aux[row][col], t)
  EK_SWEEP,        // u = acc ; o1 = sp''(z) * v * u  (second-order cotangent on z) ; o2 = sp'(z) * u (tangent of h)
  EK_VBACK,        // h = acc*scale ; col < split -> o1[row][col] = sp'(z[row][col]) * h + o1[row][col] ; else o2[row][col-split] = h
  EK_RELU_MASK,    // v = acc ; col < split -> o1[row][col] = aux[row][col] > 0 ? v : 0 ; else o2[row][col-split] = v
};

struct Epi {
  int kind = EK_STORE;
  int n_out = 0;                       // logical number of output columns (cols >= n_out are dropped)
  const float* bias = nullptr;
  float scale = 1.0f;
  float* o1 = nullptr; int ld1 = 0; int o1_off = 0;
  float* o2 = nullptr; int ld2 = 0;
  int split = 1 << 30;
  const float* z = nullptr; int ldz = 0;      // pre-activations (EK_SWEEP / EK_VBACK)
  const float* v = nullptr; int ldv = 0;      // grad-chain cotangent v_l (EK_SWEEP); ldv == 0 -> broadcast row
  float vscale = 1.0f;                        // multiplies v (the broadcast row is W_top[0,:] / scale)
  const float* aux = nullptr; int ldaux = 0;  // relu mask source / global colour
  int inv_sigmoid = 1;                        // EK_RELIGHT_TOP mode
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

CNR_HD void epi_apply(const Epi& e, long row, int col, float acc) {
  if (col >= e.n_out) return;
  switch (e.kind) {
    case EK_STORE: {
      float b = e.bias ? e.bias[col] : 0.0f;
      e.o1[row * e.ld1 + e.o1_off + col] = (acc + b) * e.scale;
    } break;
    case EK_SPLIT: {
      float b = e.bias ? e.bias[col] : 0.0f;
      float v = (acc + b) * e.scale;
      if (col < e.split) e.o1[row * e.ld1 + e.o1_off + col] = v;
      else if (e.o2) e.o2[row * e.ld2 + (col - e.split)] = v;
    } break;
    case EK_SDF_TOP: {
      float v = acc + e.bias[col];
      if (col == 0) e.o2[row] = v * e.scale;
      else e.o1[row * e.ld1 + (col - 1)] = v;
    } break;
    case EK_RELU: {
      float v = acc + e.bias[col];
      e.o1[row * e.ld1 + col] = v > 0.0f ? v : 0.0f;
    } break;
    case EK_SIGMOID: e.o1[row * e.ld1 + col] = sigmoidf_(acc + e.bias[col]); break;
    case EK_LINEAR_SIG: e.o1[row * e.ld1 + col] = acc + e.bias[col]; break;
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
        e.o2[row * e.ld2 + (col - e.split)] = h;
      }
    } break;
    default: {  // EK_RELU_MASK
      if (col < e.split) e.o1[row * e.ld1 + col] = e.aux[row * e.ldaux + col] > 0.0f ? acc : 0.0f;
      else if (e.o2) e.o2[row * e.ld2 + (col - e.split)] = acc;
    } break;
  }
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
};

}  // namespace cnr
