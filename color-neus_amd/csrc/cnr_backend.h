// Kernel launch interface between the host orchestration (cnr_plan.cpp) and the kernels.
// Implemented twice: cnr_kernels.hip (HIP, gfx950 -- the product) and cnr_kernels_emu.cpp (CPU emulation, tests only).
#pragma once
#include "cnr_common.h"
#include "cnr_views.h"

namespace cnr {

constexpr int kMaxRaySamples = 256;   // per-ray kernels stage one ray in LDS: M <= 256
constexpr int kMaxOutside = 128;      // background samples per ray (N_OUTSIDE)

struct Segment { int dst, src, len; };   // internal column [dst, dst+len) <- reference column [src, src+len)

// effective-weight preparation (weight-norm, column permutation, padding, transpose)
struct PrepWeight {
  const float* g = nullptr;   // weight_g [n] or null (plain nn.Linear)
  const float* v = nullptr;   // weight_v / weight [n][k_ref]
  const float* b = nullptr;   // bias [n]
  int n = 0, k_ref = 0;
  int nseg = 0; Segment seg[4];
  float* W = nullptr; int ldw = 0; int npad = 0;      // [npad][ldw]
  float* Wt = nullptr; int ldwt = 0; int kpad = 0;    // [kpad][ldwt]
  float* bias = nullptr;                              // [npad]
  int row_rot = 0;            // internal row i holds reference row (i + row_rot) % n  (SDF top layer: [features | sdf])
};

// dW partial reduction + weight-norm backward + un-permutation
struct FinishWeight {
  const float* partial = nullptr; int nchunk = 0; int npad = 0; int ldk = 0;   // [nchunk][npad][ldk]
  const float* colsum = nullptr;                                             // [ncolsum][npad] or null
  int ncolsum = 0;            // slots that carry column sums (the first group of a region); 0 with colsum != null means nchunk
  const float* g = nullptr; const float* v = nullptr;
  int n = 0, k_ref = 0;
  int nseg = 0; Segment seg[4];
  float* dg = nullptr; float* dv = nullptr; float* db = nullptr;   // outputs (dg null for plain Linear; dv = d weight)
  int row_rot = 0;            // see PrepWeight
  int col_hi = 1 << 30, nchunk_hi = 0;   // columns >= col_hi are summed over nchunk_hi slots (strip columns filled by be_strip_bwd)
};

struct EmbedZ {   // E[r*m + j][0..kEmb) = PE(scale * (o_r + d_r * z[r*ldz + j])), optional coarse-z generation
  const float* o; const float* d; long R; int m;
  float* z; int ldz;            // when make_z: written; else read
  int make_z;                   // 1: z = near + (far-near)*lin(j) (+ (t_rand-0.5)*2/S)          (NeuS.py:311-326)
  const float* near_; const float* far_; const float* t_rand;
  float scale; int multires;
  float* E;                     // [R*m][kEmb]
};

struct EmbedPts {  // E (and optionally AUX[.,0:3]) from explicit points or from a lattice (NeuS.py:15-24)
  const float* pts; long n;            // pts [n][3] or null -> lattice
  int res; float bmin[3], bmax[3]; long start;   // lattice point index = start + i, order x-major (x, y, z)
  float scale; int multires;
  float* E; float* AUX;
};

struct UpSample {   // NeuS.py:136-181 + ray_utils.py:123-154 (det=True)
  const float* o; const float* d; long R;
  const float* z; int ldz; const float* sdf; int lds; int n;   // current samples per ray
  int m;                                             // new samples per ray
  float inv_s;
  float* new_z;                                      // [R][m]
  const float* w_in = nullptr;                       // optional [R][n-1]: section weights given by the caller (plain sample_pdf(det=True),
                                                     // ray_utils.py:123-154, bins = z); o / d / sdf / inv_s are then unused
  const float* u_in = nullptr;                       // optional [R][m]: the uniform draws of sample_pdf(det=False) (ray_utils.py:135-136: torch.rand, drawn by the caller
                                                     // from the CPU generator like the reference); null: det=True, u_k = (k + 0.5) / m
};

struct MergeZ {     // NeuS.py:183-197
  long R; float* z; int ldz; const float* sdf_in; int lds_in; float* sdf_out; int lds_out; int n;   // z merged in place
  const float* new_z; const float* new_sdf; int m;   // new_sdf null on the last step (sdf not updated)
};

struct SamplerStep {   // one launch per up-sampling iteration: [merge of the previous iteration] + up_sample + [PE rows of the new samples]
  int do_merge; MergeZ g;          // cat_z_vals of the previous iteration's new samples (its new_z / new_sdf are read before u.new_z is rewritten)
  UpSample u;                      // on the merged row (u.n = g.n + g.m after a merge); u.w_in must be null
  int do_embed; float* E; float scale; int multires;   // E[(ray * u.m + j)][kEmb] = PE(scale * (o + d * new_z[j])) like EmbedZ
};
void be_sampler_step(const SamplerStep& p, cnr_stream s);

struct FineSetup {  // Color_NeuS.py:41-50: section midpoints, PE input of the SDF net, auxiliary inputs
  const float* o; const float* d; const float* z; long R; int M; float sample_dist;
  float scale; int multires; int multires_view;
  float* E;      // [P][kEmb]
  float* AUX;    // [P][kAux] = [p(3) g(3) PE(dir)(3+6*mv) 0...]
};

struct GradFinish { // g = scale * J_PE^T (ce0 + ce_skip)
  long P; const float* E; const float* ce0; const float* ces; float scale; int multires;
  float* grad_out;   // [P][3] (the 'gradients' output tensor)
  float* AUX;        // g copied to AUX[.,3:6]
  int neg_g_as_view; int multires_view;   // vertex colouring (NeuS.py:60): view_dirs = -g -> AUX[.,6:] = PE(-g)
  float* featx; int ldfx; int F;          // colour-net input [feat | aux | 0]: the aux row is copied to featx[.,F:ldfx)
};

struct CompositeFwd {   // Color_NeuS.py:66-123, NeuS.py:382-399
  const float* o; const float* d; const float* z; long R; int M; float sample_dist;
  const float* sdf; const float* g; const float* color; int ldcolor; const float* gcolor; int ldg;   // gcolor null for plain NeuS
  const float* variance; float cos_anneal; const float* background_rgb;
  float* color_fine; float* s_val; float* cdf_fine; float* weight_sum; float* weight_max; float* weights;
  float* inside_sphere; float* depth; float* global_color;
  float* eik_partial;    // [R][2]
  // optional per-sample copies of the network outputs for callers that composite themselves (N_OUTSIDE > 0 background mixing)
  float* sdf_s; float* color_s; float* gcolor_s;   // [P], [P][3], [P][3] or null
  const float* delta = nullptr; float* delta_ray_sum = nullptr;   // optional: [P][3] relight offsets -> [R] sums over samples and rgb
};

// inference-only early-termination compaction: samples whose compositing weight is below eps contribute < eps to the pixel,
// so the colour / relight networks are evaluated only on the kept samples (valid when the caller consumes colour / depth only)
struct PruneCount { const float* weights; long R; int M; float eps; int* counts; };
struct PruneScan { const int* counts; long R; int* offsets /*[R+1]*/; };
struct PruneGather {
  const float* weights; long R; int M; float eps; const int* offsets;
  int* idx;                                       // [kept] original point index
  const float* featx; int ldfx; float* featx_c;   // rows gathered to the compact list (featx_c null: index list only)
  const float* aux; float* aux_c;                 // [.][kAux]
  // index-list form (forward-only render, chain-fused ReLU stacks read their rows through idx): the per-sample outputs of the DROPPED samples
  // are zeroed here instead of by three memsets over the whole buffers
  float* zero_gcol = nullptr; float* zero_relit = nullptr; float* zero_delta = nullptr;   // [P][4], [P][4] or null, [P][3] or null
};
struct PruneScatter {
  long P; const int* count; const int* idx;
  const float* gcol_c; const float* relit_c; const float* delta_c;   // compact [.][4], [.][4], [.][3] (relit/delta null for plain NeuS)
  float* gcol; float* relit; float* delta;                            // full-size, pre-zeroed
};

struct ReduceEik { const float* partial; long R; float* sums /*[2]*/; float* sums_out /*[2] or null*/; float* gradient_error; };

struct CompositeBwd {
  const float* o; const float* d; const float* z; long R; int M; float sample_dist;
  const float* sdf; const float* g; const float* color; int ldcolor; const float* gcolor; int ldg;
  const float* variance; float cos_anneal; const float* background_rgb; const float* eik_sums; float sdf_scale;
  int inv_sigmoid; int has_relight;
  // upstream gradients (any may be null)
  const float* d_color_fine; const float* d_s_val; const float* d_cdf; const float* d_weight_sum; const float* d_weight_max;
  const float* d_gradients; const float* d_weights; const float* d_gradient_error; const float* d_depth;
  const float* d_global_color; const float* d_delta_relight; const float* d_delta_relight_ray /* [R] or null, see cnr_render_out_grads */;
  const float* d_sdf_s; const float* d_color_s; const float* d_gcolor_s;   // per-sample cotangents of CompositeFwd::sdf_s / color_s / gcolor_s (or null)
  // per-point cotangents
  float* ztop; int ldztop; int ztop_col;   // ztop[pt][ztop_col] = d sdf / scale (the sdf row is the LAST internal row of the top layer)
  float* gbar;                 // [P][4] d loss / d g through alpha, eikonal and the 'gradients' output
  int ldtop = kTop;            // row stride of dtop / gc_a: kTop (16: the narrow GEMM launches read 16-float rows) or 4 (packed: the streaming head kernels)
  float* dtop;                 // [P][ldtop] cotangent of the last relight layer output (pre-activation), zero padded
  float* gc_a;                 // [P][ldtop] cotangent of the global colour (post-sigmoid): direct + inverse-sigmoid path
  float* dinvs_partial;        // [R]
  float* d_rays_d;             // [R][3] (null when rays need no grad): sum_j d tc_j * g_j
  float* d_z;                  // [P][2] or null: {d loss / d z_j through depth, d loss / d dist_j through alpha} (near / far gradients, N_IMPORTANCE == 0)
};

struct ColTopBwd {  // cotangent of the colour net's last pre-activation
  long P; const float* gc_a /*[P][ldtop]*/; const float* gc_b /* [P][4], may be null */; const float* gcolor; int squeeze; float* out; /*[P][ldtop]*/
  int ldtop = kTop;
};

struct GbarFinish { // total d g, then tangent of the embedding: cbar = J_PE (scale * gbar)
  long P; const float* gbar_alpha; const float* daux_c; const float* daux_r; const float* E; float scale; int multires;
  float* gbar_total /*[P][4]*/; float* cbar /*[P][kEmb]*/;
};

struct VarianceFinish { const float* partial; long R; const float* variance; float* d_variance; };

struct RaysGradFinish {  // d rays_o / d rays_d from the point cotangents (only when rays require grad)
  long R; int M; const float* d; const float* z; float sample_dist;
  const float* pbar;      // [P][4] total cotangent of p
  const float* daux_dir_c; const float* daux_dir_r; int lddir; int multires_view;   // cotangent of PE(dir) parts (may be null)
  const float* d_rays_d_alpha;   // [R][3] from CompositeBwd
  float* d_o; float* d_d;        // may both be null (only near / far need gradients)
  // N_IMPORTANCE == 0: z_j = near + (far - near) * linspace(0,1,S)[j] (+ jitter) is differentiable w.r.t. near / far (NeuS.py:311-313)
  const float* dz_parts;         // [P][2] from CompositeBwd::d_z, or null
  float* d_near; float* d_far;   // [R] or null
};

struct PbarFinish {   // total cotangent of p: colour/relight aux inputs + SDF value path + gradient path (PE second derivative)
  long P; const float* daux_c; const float* daux_r; const float* ebar0; const float* ebars; const float* E;
  const float* ce0; const float* ces; const float* gbar_total; float scale; int multires; float* pbar;
};

// ---- chain-fused kernels (cnr_chain.hip): a tile of points goes through all layers of a chain without leaving the CU
struct PackJob {    // f16 planes [2][rows][ld] (be_split_planes) -> fragment-major planes Wf[8][ld/16][2][64][8] of the fused kernels
  const unsigned short* planes; long plane_stride; int rows; int ld; unsigned short* Wf;
};
void be_pack_frags_many(const PackJob* jobs, int count, cnr_stream s);
struct FusedLayer {
  const unsigned short* Wf;   // fragment-major f16 planes of the row-scaled weights
  const float* wsc;           // per output column: 1 / weight row scale
  const float* bias;          // per output column (readable up to column 255; columns >= N are ignored)
  int K;                      // input width, multiple of 16, <= 256
  int N;                      // output columns, <= 256
};
struct SdfValueChain {   // sdf = SDFNetwork.sdf(x) from the embedding rows E (fields.py:81-100), value only
  const float* E; long P;          // [P][kEmb]
  int nl;                          // hidden layers; the narrow top layer (sdf row) is folded into the last epilogue
  FusedLayer lay[kMaxLayers];
  int skip_mask; int emb;          // bit l: layer l takes [softplus(z_{l-1}) | e] / sqrt(2)
  const float* wtop; const float* btop; float top_scale;   // sdf row of the top layer (fp32 effective weights), its bias, sign / scale
  float* sdf_out;                  // [P]
};
bool be_sdf_value_chain(const SdfValueChain& c, cnr_stream s);   // false: not handled (backend without fused kernels / unsupported shape)
// The SAVING SDF forward (Color_NeuS.py:52-54 on the fine samples): the value chain above that also leaves behind what the backward pass
// reads -- every hidden layer's PRE-activation row z_l (the tail columns of the layer in front of a skip connection receive e: the tail fill
// of the per-layer launches), the row scales of the layer inputs, the feature rows of the top layer -- in one launch (cnr_chain_fwd.hip).
struct SdfSaveChain {
  SdfValueChain v;                 // E, P, hidden layers, skip mask, sdf row of the top layer, sdf_out
  float* Z[kMaxLayers] = {};       // [P][ldz] per hidden layer
  int ldz = 0;
  float* rs[kMaxLayers + 1] = {};  // optional [P]: power-of-two scale of the input row of layer l (l >= 1; LayerGemm::rs_out convention)
  FusedLayer top;                  // the feature rows of the top layer (K == 256, N == 256): feat = h W^T + b, no activation
  float* feat = nullptr; int ld_feat = 0;   // [P][ld_feat]
};
bool be_sdf_save_chain(const SdfSaveChain& c, cnr_stream s);      // false: not handled
// The analytic gradient chain of the SDF network (the reverse sweep that replaces SDFNetwork.gradient, fields.py:105-115) in one launch:
//   u_l = softplus'(z_l) * v_l ;  v_{l-1} = W_l^T u_l  (/ sqrt(2) and split into [hidden | embedding] parts below a skip connection), l = L-1 .. 0,
// v_{L-1} the broadcast sdf row of the top layer.  Every v_l is stored for the backward pass (second-order sweep), the row scales of u_l with it;
// the embedding cotangents go to ce0 (layer 0) and ces (skip layer).
struct SdfGradChain {
  long P = 0; int nl = 0;
  FusedLayer lay[kMaxLayers];             // per layer l: fragment-major planes of W_l^T (a row per INPUT column of layer l), wsc = per-row inverse scales,
                                          // K = round_up(layer width n_l, 16) (contraction), N = k_int(l) (outputs = the layer's input columns)
  const float* Z[kMaxLayers] = {}; int ldz = 0;   // pre-activations of the hidden layers
  const float* vrow = nullptr; float vscale = 1.0f;   // v_{L-1}[col] = vrow[col] * vscale
  float* V[kMaxLayers] = {};              // V[l - 1] <- step l (l >= 1), row stride ldz; columns >= the width of layer l - 1 are written as zeros
  int skip_mask = 0; int n_out[kMaxLayers] = {};      // n_out[l]: hidden width n_{l-1} of the layer below (output columns that go to V[l-1]); the rest of a skip layer's output is embedding
  float* ces = nullptr; int ces_off = 0;  // [P][kEmb]: embedding cotangent of the skip layer at column offset ces_off
  float* ce0 = nullptr;                   // [P][kEmb]: output of step 0 (emb live columns, the rest zero)
  int emb = 0;
  float* rs[kMaxLayers] = {};             // optional [P]: row scale of u_l (LayerGemm::rs_out convention)
};
bool be_sdf_grad_chain(const SdfGradChain& c, cnr_stream s);      // false: not handled

// The SAVING forward chains of the ReLU stacks (colour network fields.py:161-188, relight network fields.py:332-368) in ONE launch
// (cnr_chain_fwd.hip): a 128-point tile goes through colour lin0..lin(NC-2) + the rgb head, then relight in_layer + rl_mlp[0..NR-2] + the
// relight head, without the activations leaving the CU between layers; every hidden layer's ReLU output is stored once for the backward
// pass (rows transposed through LDS into full 128-byte lines), the row scales of the layer inputs with it.
constexpr int kChainSteps = 12;
struct ChainFwdStep {
  const unsigned short* Wf = nullptr;   // fragment-major f16 planes of the layer (FusedLayer::Wf layout, nkb_w k16 blocks per column block)
  const float* wsc = nullptr;           // per output column: 1 / weight row scale (256 readable)
  const float* bias = nullptr;          // per output column (256 readable)
  int nkb_w = 0;                        // ldw / 16
  int nkb_main = 0;                     // k16 blocks of the main input segment: 16 (a 256-wide row) or 1..3 (a narrow global input)
  int nkb_x = 0;                        // k16 blocks of the extra segment (W columns from 16 * nkb_main on): 0..3
  const float* in = nullptr; int ld_in = 0;   // main segment = columns [0, 16 nkb_main) of these global rows; null: the previous step's ReLU output (on chip)
  int x_src = 0;                        // extra segment: 1 = columns [16 nkb_main, 16 (nkb_main + nkb_x)) of the same global row,
                                        // 2 = the colour head's output kept on chip ([rgb | 0...], the relight y-layer's input tail)
  float* save = nullptr; int ld_save = 0;     // [P][ld_save]: ReLU output of this step (256 columns)
  float* rs_in = nullptr;               // optional [P]: power-of-two scale of this step's whole input row (LayerGemm::rs_out convention)
  int head = 0;                         // 1 / 2: the colour / relight head follows this step (end of a chain)
};
struct ChainFwdHead { const float* W = nullptr; int ldw = 0; const float* bias = nullptr; int n = 0; };   // fp32 effective weights [n <= 4][ldw >= 256]
struct ReluChainFwd {
  long P = 0; int nsteps = 0; ChainFwdStep st[kChainSteps];
  ChainFwdHead col_head; int col_squeeze = 1;           // rgb = sigmoid(.) (EK_SIGMOID) or plain (EK_LINEAR_SIG)
  float* gcol = nullptr;                                // [P][4]
  float* rgb_tail = nullptr; int ld_tail = 0;           // optional: rgb_tail[row * ld_tail + c] = rgb (c < n) or 0 (c < 16): the relight y-layer's input tail
  ChainFwdHead rel_head; int inv_sigmoid = 1;
  float* delta = nullptr;                               // [P][3] relight offsets (pre-activation)
  float* relit = nullptr;                               // [P][4] relight_apply(rgb, delta)
  int dbg = 0;                                          // ablation switches (CNR_CHAIN_FWD_DBG; wrong results): 1 no saves, 2 no MFMAs, 4 stores without the LDS pass, 8 no global stores
  // forward-only render with early-termination compaction (inference): the chains run on the first *P_dev entries of row_idx only -- chain-start
  // rows are read from, and the heads' outputs written to, point row_idx[i]; P is then the upper bound that sizes the launch.  No saves.
  const int* P_dev = nullptr; const int* row_idx = nullptr;
};
bool be_relu_chain_fwd(const ReluChainFwd& c, cnr_stream s);   // false: not handled (per-layer launches instead)
bool be_relu_chain_fwd_enabled();                               // the debugging switches that turn the chain-fused ReLU forward off are not set

void be_layer_gemm(const LayerGemm& g, cnr_stream s);
// Second-order sweep launch of a narrow-input layer (K <= 48, 256 outputs) that also forms the gradient-chain weight-gradient pair of the layer
// (u = sp'(z) v from the epilogue's side inputs, the launch's input rows) into be_sweep0_slots(g.P) slots of [256][ldk] floats (cnr_sweep0.hip).
// Backward of a narrow-input layer (<= 48 input columns, 256 outputs) in one pass over its 256-wide output cotangent X (cnr_narrow_bwd.hip):
// dW[j][c] = sum X[pt][j] Y[pt][c] and db[j] = sum X[pt][j] into be_narrow_bwd_slots(P) slots ([256][ldk] / [256] floats each), and -- with Wp --
// dx[pt][c] = sum_j X[pt][j] Wt[c][j] for c < ndx (Wp / wscale: the f16 planes and row scales of W^T, rows = input columns, ldw = 256).
// partial == nullptr: only dx (then X may be the view sp'(X) * Xb: the 39-column end of the forward gradient chain).
struct NarrowBwd {
  const float* X = nullptr; int ldx = 0;
  const float* Xb = nullptr; int ldxb = 0;   // optional: X stands for the view softplus100'(X) * Xb (launches without a weight gradient: partial == nullptr)
  const float* Y = nullptr; int ldy = 0; int ky = 0;
  long P = 0;
  const unsigned short* Wp = nullptr; long wp_stride = 0; int ldw = 0; int w_rows = 0; const float* wscale = nullptr;
  float* dx = nullptr; int lddx = 0; int ndx = 0;
  float* partial = nullptr; int ldk = 0; float* colsum = nullptr;
};
bool be_narrow_bwd_ok(const NarrowBwd& p);
int be_narrow_bwd_slots(long P);
void be_narrow_bwd(const NarrowBwd& p, cnr_stream s);
bool be_sweep0_ok(const LayerGemm& g);
int be_sweep0_slots(long P);
void be_sweep0_dw(const LayerGemm& g, float* partial, int ldk, cnr_stream s);
void be_dw_gemm(const DwGemm& g, cnr_stream s);
// Layer launch g + the single-pair weight gradient d (X[0] = the launch's input view, Y[0] = the operand its epilogue derives from its side
// inputs, see DwFuse in cnr_views.h) with d.partial / d.colsum laid out in kFdwSlots slots.  Callers test be_fdw_enabled() && fdw_shape_ok(g)
// first; the backend may still run the two parts as separate launches (same results, same slots).
// Backward of a <= 4-wide head (rgb / relight offset: nn.Linear(256, 3) after a ReLU) in ONE streaming launch instead of a K = 3 layer launch +
// a weight-gradient strip launch that re-reads the same 1 KB/point: dout[pt][k] = aux[pt][k] > 0 ? sum_j W[j][k] dtop[pt][j] : 0 (the
// EK_RELU_MASK epilogue), dW[j][k] = sum_pt dtop[pt][j] aux[pt][k], db[j] = sum_pt dtop[pt][j]; fp32 FMAs, fixed-order partial sums per slot.
struct HeadBwd {
  const float* dtop; int ldt;     // [P][ldt] cotangent of the head's output (n live columns)
  const float* aux; int ldaux;    // [P][ldaux] the head's input = ReLU output of the layer below (K live columns)
  const float* W; int ldw;        // [n][ldw] effective weights (natural orientation)
  int n; int K; long P;           // n <= 4, K <= 256 (one thread per column)
  float* dout; int ldo;           // [P][ldo] cotangent of the layer below (pre-activation)
  float* partial; float* colsum; int npad, ldk, nslots;   // [nslots][npad][ldk], [nslots][npad]
};
void be_head_bwd(const HeadBwd& p, cnr_stream s);
// Forward pass of a narrow head (<= 4 outputs) on a 256-wide activation: y[pt][j] = sum_c h[pt][c] * W[j][c], through the head's epilogue E
// (epi_apply, columns 0..15: bias, sigmoid / relight, pad-column handling).  One streaming pass; a launch of the FP32-MFMA layer kernel
// spends a 32-column tile on the 3 columns.
struct HeadFwd {
  const float* h; int ldh; long P; const int* P_dev;   // [P][ldh], 256 live columns (ldh % 4 == 0); P_dev: optional device-side row count
  const float* W; int ldw; int n;                       // [n][ldw] effective weights
  Epi E;
};
void be_head_fwd(const HeadFwd& p, cnr_stream s);
// The few input columns beyond 256 of a 256-wide layer (relight y-layer: + rgb, colour layer 0: + p, g) in the backward pass, one streaming
// pass over the layer's output cotangent instead of a narrow layer launch plus a weight-gradient strip launch:
//   tail[pt][j]                    = tail_scale * sum_n dout[pt][n] * Wt[256 + j][n]          (cotangent of input column 256 + j)
//   partial[slot][n][256 + j]      = sum over the slot's points of dout[pt][n] * y[pt][j]     (weight-gradient strip; zero for nt <= j < ldk - 256)
struct StripBwd {
  const float* dout; int ldo; long P;   // [P][ldo], 256 live columns (ldo % 4 == 0)
  int nt;                               // extra input columns, 1..8
  const float* Wt; int ldwt;            // transposed fp32 weights [k][ldwt]: row 256 + j is used
  float* tail; int ldt; float tail_scale;   // [P][ldt] (null: cotangent not wanted)
  const float* y; int ldy;              // [P][ldy]: column j = input column 256 + j of the layer
  float* partial; int npad, ldk, nslots;    // [nslots][npad][ldk]
};
void be_strip_bwd(const StripBwd& p, cnr_stream s);
bool be_fdw_enabled();
bool be_fdw_xrow();          // DwFuse::xrow_mode / LayerGemm::k_extra are available in be_layer_dw_gemm
void be_layer_dw_gemm(const LayerGemm& g, const DwGemm& d, const DwFuse& f, cnr_stream s);
void be_prep_weight(const PrepWeight& p, cnr_stream s);
// fp32 matrix [rows][ld] -> two f16 planes of the row-scaled matrix (x * 2^e = hi + lo, 22 significand bits) + 1/2^e per row,
// for the weight-stationary f16-split GEMM (cnr_gemm.hip)
void be_split_planes(const float* src, int rows, int ld, unsigned short* planes, float* inv_scale, cnr_stream s);
// all layers of a model in two launches (the per-layer launches are latency-bound: 57 launches of a few microseconds each)
struct SplitJob { const float* src; int rows; int ld; unsigned short* planes; float* inv_scale; };
void be_prep_weights(const PrepWeight* p, int count, cnr_stream s);
void be_finish_weights(const FinishWeight* f, int count, cnr_stream s);   // every layer keeps its own partial buffer until this launch
void be_split_planes_many(const SplitJob* jobs, int count, cnr_stream s);
void be_finish_weight(const FinishWeight& p, cnr_stream s);
void be_embed_z(const EmbedZ& p, cnr_stream s);
void be_embed_pts(const EmbedPts& p, cnr_stream s);
void be_upsample(const UpSample& p, cnr_stream s);
void be_merge(const MergeZ& p, cnr_stream s);
void be_fine_setup(const FineSetup& p, cnr_stream s);
void be_grad_finish(const GradFinish& p, cnr_stream s);
void be_composite_fwd(const CompositeFwd& p, cnr_stream s);
void be_reduce_eik(const ReduceEik& p, cnr_stream s);
void be_prune_count(const PruneCount& p, cnr_stream s);
void be_prune_scan(const PruneScan& p, cnr_stream s);
void be_prune_gather(const PruneGather& p, cnr_stream s);
void be_prune_scatter(const PruneScatter& p, cnr_stream s);
void be_composite_bwd(const CompositeBwd& p, cnr_stream s);
void be_coltop_bwd(const ColTopBwd& p, cnr_stream s);
void be_gbar_finish(const GbarFinish& p, cnr_stream s);
void be_variance_finish(const VarianceFinish& p, cnr_stream s);
void be_pbar_finish(const PbarFinish& p, cnr_stream s);
void be_rays_grad_finish(const RaysGradFinish& p, cnr_stream s);
void be_memset_zero(void* p, size_t bytes, cnr_stream s);

// training loss (include/colorneus_render.h, cnr_loss_*): partial sums [nblk][4] -> sums[4]; gradients w.r.t. the renderer outputs
struct LossArgs {
  const float* color; const float* wsum; const float* drel; const float* gt; const float* mask;
  long R; int M; int rgb_l1; int include_mask;
  int drel_per_ray = 0;   // 1: drel is [R], the per-ray sums over samples and rgb (CompositeFwd::delta_ray_sum)
};
constexpr int kLossBlocks = 256;
void be_loss_sums(const LossArgs& a, float* partial /* [kLossBlocks][4] */, float* sums /* [4] */, cnr_stream s);
void be_loss_grads(const LossArgs& a, const float* coef, float* d_color, float* d_wsum, float* d_drel, cnr_stream s);
struct LossScalars {   // see cnr_loss_combine / cnr_loss_coef in the ABI header; the constants are formed in double on the host and rounded once
  float lf, le, lm, lr;                 // lambdas
  float Rg, den_rgb, den_rel;           // n_rays_global, 3 Rg, 3 Rg M
  float c_rgb, c_bce, c_rel;            // lambda_fine * (1 | 2) / (3 Rg), lambda_mask / Rg, lambda_relight * 2 / (3 Rg M)
  int use_mask, use_relight;
};
void be_loss_combine(const LossScalars& c, const float* sums, const float* gerr, float* out6, cnr_stream s);
void be_loss_coef(const LossScalars& c, const float* g_loss, const float* mean_rel, float* coef4, cnr_stream s);
// the forward side (partial sums, their fold, the scalar tail) and the backward side (coefficients + element-wise gradients) as ONE launch each
// (ticket: a zero-initialised 4-byte completion counter in the caller's scratch, left zero)
void be_loss_forward(const LossArgs& a, float* partial, unsigned* ticket, const LossScalars& c, const float* gerr, float* sums, float* out6, cnr_stream s);
void be_loss_backward(const LossArgs& a, const LossScalars& c, const float* g_loss, const float* mean_rel, const float* eik_factor /* or null */, float* coef4,
                      float* d_color, float* d_wsum, float* d_drel_ray /* [R] or null */, cnr_stream s);
// ray-sharded runs: this rank's statistics for the one all-reduce of the objective (one launch), and the scalar tail on the reduced statistics
void be_loss_shard_stats(const LossArgs& a, float* partial, unsigned* ticket, const float* eik_sums /* [2] */, float* stats8, cnr_stream s);
void be_loss_shard_combine(const LossScalars& c, const float* stats8, float* out8, cnr_stream s);
// per-parameter gradient clip + Adam over up to kAdamBatch tensors per launch (clip_gradient + torch.optim.Adam, net_utils.py:174-184, :88)
constexpr int kAdamBatch = 64;
constexpr int kAdamChunk = 4096;   // elements per workgroup: a tensor is cut into ceil(n / kAdamChunk) chunks
struct AdamTensor { float* w; const float* g; long off; long n; int chunk0; };   // off: offset of this tensor's moments in the flat m / v buffers; chunk0: its first chunk
struct AdamArgs {
  int count; AdamTensor t[kAdamBatch];
  int nchunks; float* partial;         // [nchunks] per-chunk sums of squared gradient entries (scratch)
  float* m; float* v;                  // flat exp_avg / exp_avg_sq
  float lr, beta1, beta2, eps, max_norm;   // max_norm <= 0: no clipping
  float bc1, bc2_sqrt;                 // 1 - beta1^step, sqrt(1 - beta2^step)
  const float* hyper;                  // device [3] = {lr, bc1, bc2_sqrt} or null: read instead of the three members above (graph replay)
};
// one tensor's update; norm2 = the tensor's sum of squared gradient entries (fixed-order reduction by the caller)
CNR_HD float adam_clip_coef(float norm2, float max_norm) {
  if (!(max_norm > 0.0f)) return 1.0f;
  const float c = max_norm / (sqrtf(norm2) + 1e-6f);    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
  return c < 1.0f ? c : 1.0f;
}
CNR_HD void adam_update1(const AdamArgs& a, float* w, float g, float* m, float* v) {
  const float m1 = *m + (g - *m) * (1.0f - a.beta1);               // exp_avg.lerp_(grad, 1 - beta1)
  const float v1 = *v * a.beta2 + (1.0f - a.beta2) * g * g;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  *m = m1; *v = v1;
  const float lr = a.hyper ? a.hyper[0] : a.lr, bc1 = a.hyper ? a.hyper[1] : a.bc1, bc2_sqrt = a.hyper ? a.hyper[2] : a.bc2_sqrt;
  const float denom = sqrtf(v1) / bc2_sqrt + a.eps;
  *w = *w - (lr / bc1) * (m1 / denom);                         // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
}
void be_clip_adam(const AdamArgs& a, cnr_stream s);   // a.partial must hold a.nchunks floats

// ---- ray generation for the selected pixels (the producer right in front of the path): get_rays_multicam / get_rays_at
// (lib/models/tools/ray_utils.py:16-119) with the trainer's origin / radius normalisation and near_far_from_sphere
// (NeuS_Trainer.py:117-119, ray_utils.py:7-13) folded in, and its backward to the camera poses and the focal lengths
struct GenRays {
  const long* pix_idx; long n;          // flat index over (camera, row, column); null: pixel i of camera 0 (get_rays_at)
  const float* c2w; int n_cams;         // [n_cams][4][4]
  const float* focal;                   // [2] (device)
  int H, W, normalize, opengl;
  const float* image; const float* mask;    // [n_cams][H][W][3] / [n_cams][H][W] or null
  const float* origin; float radius;        // rays_o = (rays_o - origin) / radius when origin != null
  float* rays_o; float* rays_d; float* rgb; float* mask_sel; float* near_; float* far_;   // any of rgb / mask_sel / near_ / far_ may be null
  int* bad_count = nullptr;             // device counter or null: +1 per pixel index outside [0, n_cams * H * W) (an error flag, not arithmetic)
};
struct GenRaysBwd {
  GenRays f;                            // the forward arguments (outputs unused)
  const float* d_rays_o; const float* d_rays_d; const float* d_near; const float* d_far;   // upstream gradients (d_near / d_far may be null)
  float* d_c2w;                         // [n_cams][4][4] (bottom row zero)
  float* d_focal_partial;               // [n_cams][2] per-camera contributions, folded in camera order into d_focal
  float* d_focal;                       // [2]
};
void be_gen_rays(const GenRays& p, cnr_stream s);
void be_gen_rays_bwd(const GenRaysBwd& p, cnr_stream s);

// ---- iso-surface extraction on the device-resident SDF lattice (replaces the CPU mcubes.marching_cubes call of extract_geometry,
// NeuS.py:31-40): cell classification + edge ownership, two exclusive scans, vertex / triangle emission
struct McVolume {
  const float* u; int res; float thr;      // u[x][y][z], surface where u crosses thr, "inside" = u > thr
  unsigned char* flags;                    // [res^3] bit a: the edge from this voxel along +axis a crosses the level
  int* counts;                             // [res^3][2] -> exclusive offsets {vertex, triangle} after mc_scan
  int* block_sums;                         // [nblocks][2]
  int* totals;                             // [2] device: number of vertices, number of triangles
};
constexpr int kMcScanBlock = 2048;
void be_mc_count(const McVolume& v, cnr_stream s);    // flags + per-voxel {vertex, triangle} counts, scanned into offsets; totals written
void be_mc_emit(const McVolume& v, const float* bmin, const float* bmax, float* verts, int* tris, cnr_stream s);
// shared per-voxel logic (host + device)
CNR_HD int mc_cell(const float* u, int res, float thr, int x, int y, int z, unsigned char* flags_out) {   // returns the cube index or -1 (no cell)
  const long r2 = (long)res * res, v = ((long)x * res + y) * res + z;
  const bool f0 = u[v] > thr;
  unsigned char fl = 0;
  if (x + 1 < res && (u[v + r2] > thr) != f0) fl |= 1;
  if (y + 1 < res && (u[v + res] > thr) != f0) fl |= 2;
  if (z + 1 < res && (u[v + 1] > thr) != f0) fl |= 4;
  *flags_out = fl;
  if (x + 1 >= res || y + 1 >= res || z + 1 >= res) return -1;
  int idx = 0;
  for (int c = 0; c < 8; ++c) {
    const long vc = v + (c & 1) * r2 + ((c >> 1) & 1) * res + ((c >> 2) & 1);
    idx |= (u[vc] > thr ? 1 : 0) << c;
  }
  return idx;
}
CNR_HD int mc_popcount3(unsigned x) { return (int)((x & 1) + ((x >> 1) & 1) + ((x >> 2) & 1)); }

// ---- N_OUTSIDE > 0: the NeRF++ background of NeuS (NeuS.py:95-134, 313-369; NeRF, fields.py:192-274) -----------------------------------
// No shipped configuration enables it; these kernels favour plain code over speed (one thread per ray / per point).
struct OutsideZ {       // background sample positions (NeuS.py:315-338) merged with the sorted foreground samples (NeuS.py:353-355)
  long R; int M, n_out, n_samples;
  const float* far_; const float* t_rand /* [R][n_out] uniform draws (NeuS.py:335) or null: no perturbation */; const float* z /* [R][M], ascending */;
  float* z_feed;        // [R][M + n_out] ascending
  int* src;             // [R][M + n_out]: j >= 0: z[j];  -1 - k: background sample k
};
struct OutsideZBwd {    // d far (z_out[k] = far / zz[n_out - 1 - k] + 1 / n_samples) and, for the copies of z in z_feed, d z
  long R; int M, n_out, n_samples; const float* t_rand; const int* src; const float* d_z_feed;
  float* d_far;         // [R]
  float* d_z;           // [R][M] or null
};
struct BgEmbed {        // render_core_outside's inputs (NeuS.py:102-115): section length, mid point, [p / r, 1 / r], PE-multires of it, PE-multires_view of the direction
  const float* o; const float* d; const float* z_feed; long R; int MF; float sample_dist; int multires, multires_view;
  float* E; int lde;               // [n][lde]: PE row (4 + 8 multires live columns, zero padded)
  float* XH; int ldxh, xh_off;     // optional second copy of the PE row: XH[pt][xh_off + c] (the skip layer's input tail, zero padded up to ldxh)
  float* FV; int ldfv, fv_off;     // PE of the view direction: FV[pt][fv_off + c] (the view layer's input tail, zero padded up to ldfv)
  float* dist;                     // [n] section lengths
};
struct BgAlpha {        // alpha = 1 - exp(-softplus(density) dist) (NeuS.py:119-120)
  long n; const float* density; const float* dist; float* alpha;
};
struct BgHeadsBwd {     // cotangents of the two heads' pre-activations: d density (through alpha) and d rgb_pre (through the sigmoid), d dist
  long n; const float* density; const float* dist; const float* rgb /* [n][3] post-sigmoid */; const float* d_alpha; const float* d_rgb /* [n][3] */;
  float* d_density /* [n][ldd], column 0 live, the rest zero */; int ldd; float* d_rgb_pre /* [n][ldr], 3 live */; int ldr; float* d_dist /* [n] */;
};
struct BgJoin {         // dZ[pt][c] = H[pt][c] > 0 ? T[pt][c] + d_density[pt] * w_alpha[c] : 0: the last hidden layer feeds the feature layer AND the density head
  long n; int W; const float* T; const float* H; const float* d_density; int ldd; const float* w_alpha; float* dZ;
};
struct BgEmbedBwd {     // per point: cotangent of the sample point p and of the view direction from the PE cotangents
  const float* o; const float* d; const float* z_feed; long R; int MF; float sample_dist; int multires, multires_view;
  const float* dE0; int lde0; const float* dE1; int lde1 /* second PE cotangent (skip layer) or null */; const float* dVE; int ldve;
  float* dp;            // [n][8]: {dp.x, dp.y, dp.z, dview.x, dview.y, dview.z, 0, 0}
};
struct BgRaysBwd {      // per ray: d rays_o, d rays_d, d z_feed from the per-point cotangents
  const float* d; const float* z_feed; long R; int MF; float sample_dist; const float* dp; const float* d_dist;
  float* d_o; float* d_d; float* d_z_feed;   // [R][3], [R][3], [R][MF] (d_z_feed is ADDED to: the compositor's depth term lands there first)
};
struct CompositeBg {    // render_core with a background (NeuS.py:236-292, Color_NeuS.py:66-138): S-density alpha, inside / outside mixing, compositing over M + n_out samples
  const float* o; const float* d; const float* z; const float* z_feed; long R; int M, MF; float sample_dist;
  const float* sdf; const float* g /* [R][M][3] */; const float* color /* [R][M][3] */; const float* gcolor /* [R][M][3] or null */;
  const float* bg_alpha /* [R][MF] */; const float* bg_color /* [R][MF][3] */;
  const float* variance; float cos_anneal; const float* background_rgb;
  float* color_fine; float* s_val; float* cdf_fine; float* weight_sum; float* weight_max; float* weights /* [R][MF] */; float* inside_sphere;
  float* depth; float* global_color; float* eik_partial /* [R][2] */;
};
struct CompositeBgBwd {
  CompositeBg f;        // the forward arguments (outputs unused except weights)
  const float* d_color_fine; const float* d_s_val; const float* d_cdf; const float* d_weight_sum; const float* d_weight_max; const float* d_weights;
  const float* d_gradient_error; const float* d_depth; const float* d_global_color; const float* d_gradients /* [R][M][3] or null */;
  const float* eik_sums;   // [2] of the forward pass
  float* d_sdf; float* d_g; float* d_color; float* d_gcolor; float* d_bg_alpha; float* d_bg_color;
  float* d_inv_s_partial /* [R] */; float* d_rays_d /* [R][3] */; float* d_z /* [R][M] */; float* d_z_feed /* [R][MF] */;
};
void be_outside_z(const OutsideZ& p, cnr_stream s);
void be_outside_z_bwd(const OutsideZBwd& p, cnr_stream s);
void be_bg_embed(const BgEmbed& p, cnr_stream s);
void be_bg_alpha(const BgAlpha& p, cnr_stream s);
void be_bg_heads_bwd(const BgHeadsBwd& p, cnr_stream s);
void be_bg_join(const BgJoin& p, cnr_stream s);
void be_bg_embed_bwd(const BgEmbedBwd& p, cnr_stream s);
void be_bg_rays_bwd(const BgRaysBwd& p, cnr_stream s);
void be_composite_bg(const CompositeBg& p, cnr_stream s);
void be_composite_bg_bwd(const CompositeBgBwd& p, cnr_stream s);

// p[row][c] = 0 for c in [c0, c1), row < rows: zero the pad columns a GEMM reads without touching the rest of a wide buffer
void be_zero_cols(float* p, int ld, int c0, int c1, long rows, cnr_stream s);
// dst[row][c] = src[row][c] for c < ncols (two row-major matrices with different row strides): compact <-> padded operand buffers
void be_copy_cols(float* dst, int ld_dst, const float* src, int ld_src, int ncols, long rows, cnr_stream s);
void be_grid_points(float* pts /*unused*/, cnr_stream s);
struct KernelTiming { char name[32]; int kind; int nt; long P; int N, K, pairs; float ms; double bytes; };
void be_timing_enable(int on);
int be_timing_collect(KernelTiming* out, int max_records);   // synchronises the recorded events, returns #records, resets
const char* be_name();
int be_check_last_error(char* msg, size_t n);   // 0 ok
// Named ranges for profilers (rocprofv3 --marker-trace): roctxRangePush / Pop of librocprofiler-sdk-roctx / libroctx64, loaded on first use
// when CNR_ROCTX=1 (no link dependency; without the variable or the library these are no-ops)
void be_range_push(const char* name);
void be_range_pop();
struct RangeScope { explicit RangeScope(const char* name) { be_range_push(name); } ~RangeScope() { be_range_pop(); } };

}  // namespace cnr
