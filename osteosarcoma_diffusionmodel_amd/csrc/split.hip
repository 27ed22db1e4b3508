// split.hip -- the "bf16x3" sampling engine: the denoiser forward and the reverse chain with every GEMM on the bf16 matrix pipe at
// fp32 accuracy (gemm_bf3.h), opt-in through osd_set_option("precision", 1).  Default precision (0) never reaches this file.
//
// Per-layer launches over row chunks, the per-layer fp32 engine's scheme (api.hip: chain_chunk) -- NOT a persistent chain kernel:
// a 16-k stage of both operands is 24 KiB for 0.52 MFLOP, 1.5x the fp32 kernels' bytes at 2.7x their rate, so the operand stream
// (measured: 9 TB/s of LDS-DMA at 196 TFLOP/s effective, tools/probes/split_probe.hip) only holds up when it hits in L2, i.e. when
// all workgroups of an XCD share one layer's weights (1.5 MB of planes for a 512 x 512 layer); the chain kernels' 64 workgroups per
// XCD sit at 64 different layers (L2 hit rate 0.38-0.69).  Activations travel between launches as planes
// buffers (6 B per element); the chain state x stays fp32 in the caller's rows and is mirrored as planes for input_proj.
#include <algorithm>
#include <vector>
#include "gemm_bf3.h"
#include "b3_pack.h"
#include "handle.h"
#include "kernels.h"
#include "fwd.h"
#include "launch.h"
#include "split.h"

namespace osd {

// eval mode, trunk widths whose GroupNorm groups are 32 / 64 channels (the fp32 chain kernels' architectures)
bool split_supported(const Arch& a) {
  if (a.H0 != 256 && a.H0 != 512) return false;
  for (int c : a.block_out)
    if (c != 256 && c != 512) return false;
  return true;
}

struct SplitPlan {
  uint4* w = nullptr;                 // all weight planes, one allocation
  int64_t units = 0;
  uint4* w_in = nullptr; uint4* w_out = nullptr;
  std::vector<uint4*> w_layer;        // 2 per block, execution order
};

template <class Epi, int NKB>
struct B3Registrar {
  B3Registrar() { kernel_registry().push_back({reinterpret_cast<const void*>(gemm_bf3_kernel<Epi, B3_LD, NKB>), B3_LDS_BYTES}); }
  static B3Registrar instance;
};
template <class Epi, int NKB> B3Registrar<Epi, NKB> B3Registrar<Epi, NKB>::instance;

template <class Epi, int NKB = 0>
static hipError_t launch_b3(hipStream_t s, const Bf3Args& g, const typename Epi::Args& ea) {
  (void)&B3Registrar<Epi, NKB>::instance;
  if (g.F <= 0 || g.P <= 0) return hipSuccess;
  const int nft = (g.F + B3_ROWS - 1) / B3_ROWS, npt = (g.P + B3_ROWS - 1) / B3_ROWS;
  const int grid = ((npt + 7) / 8) * 8 * nft;
  hipLaunchKernelGGL((gemm_bf3_kernel<Epi, B3_LD, NKB>), dim3(grid), dim3(NTHREADS + 64 * B3_LD), B3_LDS_BYTES, s, g, ea);
  return hipGetLastError();
}
// output_proj + posterior: the unrolled kernels (the normals are drawn inside the K loop) for 256 / 512 deep reductions
static hipError_t launch_post(hipStream_t s, const Bf3Args& g, EpiB3Post::Args ea) {
  ea.x_tile = (ea.ldx % 4 == 0 && g.F % 4 == 0 && g.F >= 4 && (reinterpret_cast<uintptr_t>(ea.x) & 15) == 0) ? 1 : 0;
  if (g.nkb == 16) return launch_b3<EpiB3Post, 16>(s, g, ea);
  if (g.nkb == 32) return launch_b3<EpiB3Post, 32>(s, g, ea);
  return launch_b3<EpiB3Post, 0>(s, g, ea);
}

static hipError_t launch_pack(hipStream_t s, const float* src, int ld, int64_t R, int K, uint4* dst) {
  const int nkb = b3_nkb(K);
  const long long total = b3_tiles(R) * nkb * 256;
  if (total <= 0) return hipSuccess;
  const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(k_b3_pack, dim3(grid), dim3(256), 0, s, src, ld, (long long)R, K, dst, nkb, total);
  return hipGetLastError();
}

void split_free(osd_handle* h) {
  SplitPlan* p = static_cast<SplitPlan*>(h->split_plan);
  if (!p) return;
  if (p->w) { hipError_t e = hipFree(p->w); (void)e; }
  delete p;
  h->split_plan = nullptr;
  h->split_valid = false;
}

// weight planes follow the current parameters: rebuilt by the first split-precision call after anything changed them
static int split_pack_weights(osd_handle* h, hipStream_t s) {
  const Arch& a = h->arch;
  SplitPlan* p = static_cast<SplitPlan*>(h->split_plan);
  if (!p) {
    p = new (std::nothrow) SplitPlan();
    if (!p) { set_error("out of host memory"); return OSD_ENOMEM; }
    h->split_plan = p;
    int64_t units = b3_units(a.H0, a.D) + b3_units(a.D, a.block_out[a.n_blocks - 1]);
    for (const LayerDesc& l : a.layers) units += b3_units(l.N, l.K1 + l.K2);
    void* q = nullptr;
    if (hipMalloc(&q, (size_t)units * 16) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %lld bytes failed", (long long)units * 16); return OSD_ENOMEM; }
    p->w = (uint4*)q; p->units = units;
    int64_t off = 0;
    p->w_in = p->w + off; off += b3_units(a.H0, a.D);
    p->w_layer.clear();
    for (const LayerDesc& l : a.layers) { p->w_layer.push_back(p->w + off); off += b3_units(l.N, l.K1 + l.K2); }
    p->w_out = p->w + off;
    h->split_valid = false;
  }
  if (h->split_valid) return OSD_OK;
  const ParamMap& pm = a.pm;
  OSD_HIP(launch_pack(s, h->params[pm.in_w], a.D, a.H0, a.D, p->w_in));
  for (size_t i = 0; i < a.layers.size(); ++i) {
    const LayerDesc& l = a.layers[i];
    OSD_HIP(launch_pack(s, h->params[l.w], l.K1 + l.K2, l.N, l.K1 + l.K2, p->w_layer[i]));
  }
  const int hl = a.block_out[a.n_blocks - 1];
  OSD_HIP(launch_pack(s, h->params[pm.out_w], hl, a.D, hl, p->w_out));
  h->split_valid = true;
  return OSD_OK;
}

// workspace of one row chunk: the conditioning branch in fp32 (run_cond), everything else as planes
struct SplitWs {
  FwdWs cond;                      // ce1, ce2, cproj only
  uint4* xpl; uint4* h0;
  std::vector<uint4*> mid, out;
};
static int64_t up64(int64_t v) { return (v + 63) / 64 * 64; }
static int64_t carve_split(const Arch& a, float* base, int64_t n, SplitWs* ws) {
  int64_t off = 0;
  auto take = [&](int64_t floats) { float* q = base ? base + off : nullptr; off += up64(floats); return q; };
  auto take_pl = [&](int K) { return reinterpret_cast<uint4*>(take(b3_units(n, K) * 4)); };
  ws->cond.ce1 = take(n * 64); ws->cond.ce2 = take(n * 64); ws->cond.cproj = take(b3_tiles(n) * B3_ROWS * a.H0);
  ws->xpl = take_pl(a.D);
  ws->h0 = take_pl(a.H0);
  ws->mid.resize(a.n_blocks); ws->out.resize(a.n_blocks);
  for (int b = 0; b < a.n_blocks; ++b) { ws->mid[b] = take_pl(a.block_out[b]); ws->out[b] = take_pl(a.block_out[b]); }
  return off;
}

struct SplitStep {
  const int* t_index; const int* t_dev; int t_imm;
};

// input_proj + blocks on planes; result in ws.out[n_blocks - 1]
static int split_trunk(osd_handle* h, hipStream_t s, const SplitWs& ws, int64_t n, const SplitStep& st) {
  const Arch& a = h->arch;
  const ParamMap& pm = a.pm;
  const SplitPlan& p = *static_cast<SplitPlan*>(h->split_plan);
  {
    const int nkb = b3_nkb(a.D);
    Bf3Args g{p.w_in, nkb, ws.xpl, nkb, nullptr, 0, a.H0, (int)n, nullptr};
    EpiB3Input::Args ea{h->params[pm.in_b], h->d_temb, a.H0, st.t_index, st.t_dev, st.t_imm, ws.cond.cproj, a.H0, B3Out{ws.h0, b3_nkb(a.H0)}};
    OSD_HIP(launch_b3<EpiB3Input>(s, g, ea));
  }
  const uint4* cur = ws.h0;
  int cur_w = a.H0;
  for (int b = 0; b < a.n_blocks; ++b) {
    const LayerDesc& l1 = a.layers[2 * b];
    const LayerDesc& l2 = a.layers[2 * b + 1];
    Bf3Args g{p.w_layer[2 * b], b3_nkb(l1.K1 + l1.K2), cur, b3_nkb(l1.K1), nullptr, 0, l1.N, (int)n, nullptr};
    if (l1.K2 > 0) {
      const int skip_block = a.n_enc - 1 - (b - a.n_enc - 1);   // LIFO: decoder j pops encoder n_enc-1-j (api.hip: run_trunk)
      g.B1 = ws.out[skip_block]; g.nkb1 = b3_nkb(a.block_out[skip_block]);
    }
    (void)cur_w;
    const B3Out o1{ws.mid[b], b3_nkb(l1.N)}, o2{ws.out[b], b3_nkb(l2.N)};
    if (l1.gw == 64) OSD_HIP(launch_b3<EpiB3Gn<64>>(s, g, EpiB3Gn<64>::Args{h->params[l1.b], h->params[l1.gamma], h->params[l1.beta], o1}));
    else OSD_HIP(launch_b3<EpiB3Gn<32>>(s, g, EpiB3Gn<32>::Args{h->params[l1.b], h->params[l1.gamma], h->params[l1.beta], o1}));
    Bf3Args g2{p.w_layer[2 * b + 1], b3_nkb(l2.K1), ws.mid[b], b3_nkb(l2.K1), nullptr, 0, l2.N, (int)n, nullptr};
    if (l2.gw == 64) OSD_HIP(launch_b3<EpiB3Gn<64>>(s, g2, EpiB3Gn<64>::Args{h->params[l2.b], h->params[l2.gamma], h->params[l2.beta], o2}));
    else OSD_HIP(launch_b3<EpiB3Gn<32>>(s, g2, EpiB3Gn<32>::Args{h->params[l2.b], h->params[l2.gamma], h->params[l2.beta], o2}));
    cur = ws.out[b];
    cur_w = l2.N;
  }
  return OSD_OK;
}

static Bf3Args split_out_args(osd_handle* h, const SplitWs& ws, int64_t n) {
  const Arch& a = h->arch;
  const SplitPlan& p = *static_cast<SplitPlan*>(h->split_plan);
  const int hl = a.block_out[a.n_blocks - 1];
  return Bf3Args{p.w_out, b3_nkb(hl), ws.out[a.n_blocks - 1], b3_nkb(hl), nullptr, 0, a.D, (int)n, nullptr};
}

// rows beyond n in the last tile of cproj: the input_proj epilogue clamps its row index, nothing to pad
int split_denoiser_forward(osd_handle* h, const float* x, const int* t_idx, int32_t t_all, const float* cond, int64_t n, float* eps) {
  const Arch& a = h->arch;
  hipStream_t s = h->stream;
  OSD_TRY(split_pack_weights(h, s));
  SplitWs ws;
  const int64_t need = carve_split(a, nullptr, n, &ws);
  OSD_TRY(ensure_arena(&h->main, need));
  carve_split(a, h->main.arena, n, &ws);
  OSD_TRY(run_cond(h, s, cond, n, ws.cond));
  OSD_HIP(launch_pack(s, x, a.D, n, a.D, ws.xpl));
  OSD_TRY(split_trunk(h, s, ws, n, SplitStep{t_idx, nullptr, t_all}));
  OSD_HIP(launch_b3<EpiB3Bias>(s, split_out_args(h, ws, n), EpiB3Bias::Args{h->params[a.pm.out_b], eps, a.D}));
  h->last_precision = 1;
  return OSD_OK;
}

int split_p_sample_step(osd_handle* h, const float* x_t, int32_t t, const float* cond, const float* z, int64_t n, uint64_t seed, int64_t row_offset,
                        float* x_out) {
  const Arch& a = h->arch;
  hipStream_t s = h->stream;
  OSD_TRY(split_pack_weights(h, s));
  SplitWs ws;
  const int64_t need = carve_split(a, nullptr, n, &ws);
  OSD_TRY(ensure_arena(&h->main, need));
  carve_split(a, h->main.arena, n, &ws);
  OSD_TRY(run_cond(h, s, cond, n, ws.cond));
  OSD_HIP(launch_pack(s, x_t, a.D, n, a.D, ws.xpl));
  if (x_out != x_t) OSD_HIP(hipMemcpyAsync(x_out, x_t, (size_t)n * a.D * 4, hipMemcpyDeviceToDevice, s));
  OSD_TRY(split_trunk(h, s, ws, n, SplitStep{nullptr, nullptr, t}));
  EpiB3Post::Args ea{};
  ea.bias = h->params[a.pm.out_b]; ea.x = x_out; ea.ldx = a.D; ea.coef = h->d_coef; ea.t_dev = nullptr; ea.t_imm = t;
  ea.z = z; ea.ldzz = a.D; ea.z_step_stride = 0; ea.t_first = t; ea.seed = seed; ea.row_offset = (uint32_t)row_offset;
  ea.mut_mask = nullptr; ea.mutation_dim = 0; ea.o = B3Out{ws.xpl, b3_nkb(a.D)};
  OSD_HIP(launch_post(s, split_out_args(h, ws, n), ea));
  h->last_precision = 1;
  return OSD_OK;
}

static int release_graph(Slot& sl) {
  if (!sl.exec && !sl.graph) return OSD_OK;
  OSD_HIP(hipStreamSynchronize(sl.stream));
  if (sl.exec) OSD_HIP(hipGraphExecDestroy(sl.exec));
  if (sl.graph) OSD_HIP(hipGraphDestroy(sl.graph));
  sl.exec = nullptr;
  sl.graph = nullptr;
  return OSD_OK;
}

// One chunk of the reverse chain on one slot: rows [r0, r0 + m).  The fp32 state lives in the caller's output rows.
int split_chain_chunk(osd_handle* h, Slot& sl, const float* cond, int64_t n_total, int64_t r0, int64_t m, const float* x_T, const float* noises,
                      uint64_t seed, int64_t row_offset, float* x_out, float* mut_mask_out, int flags) {
  const Arch& a = h->arch;
  const int D = a.D, T = a.T;
  hipStream_t s = sl.stream;
  OSD_TRY(release_graph(sl));
  SplitWs ws;
  const int64_t need = carve_split(a, nullptr, m, &ws);
  OSD_TRY(ensure_arena(&sl, need));
  carve_split(a, sl.arena, m, &ws);
  float* x = x_out + r0 * D;
  const uint32_t roff = (uint32_t)(row_offset + r0);
  OSD_TRY(run_cond(h, s, cond + r0 * a.cond_dim, m, ws.cond));       // loop-invariant in eval mode: hoisted (api.hip: chain_chunk)
  if (x_T) { if (x_T + r0 * D != x) OSD_HIP(launch_copy2d(s, x_T + r0 * D, D, x, D, m, D)); }
  else OSD_HIP(launch_fill_randn(s, x, D, m, D, seed, roff, (uint32_t)T, TAG_POSTERIOR));
  OSD_HIP(launch_pack(s, x, D, m, D, ws.xpl));
  OSD_HIP(launch_set_int(s, sl.t_dev, T - 1));

  auto enqueue_step = [&](void) -> int {
    OSD_TRY(split_trunk(h, s, ws, m, SplitStep{nullptr, sl.t_dev, 0}));
    EpiB3Post::Args ea{};
    ea.bias = h->params[a.pm.out_b]; ea.x = x; ea.ldx = D; ea.coef = h->d_coef; ea.t_dev = sl.t_dev; ea.t_imm = 0;
    ea.z = noises ? noises + r0 * D : nullptr; ea.ldzz = D; ea.z_step_stride = (long long)n_total * D; ea.t_first = T - 1;
    ea.seed = seed; ea.row_offset = roff;
    ea.mut_mask = mut_mask_out ? mut_mask_out + r0 * h->cfg.mutation_dim : nullptr; ea.mutation_dim = h->cfg.mutation_dim;
    ea.o = B3Out{ws.xpl, b3_nkb(D)};
    OSD_HIP(launch_post(s, split_out_args(h, ws, m), ea));
    OSD_HIP(launch_add_int(s, sl.t_dev, -1));
    return OSD_OK;
  };
  if (flags & OSD_F_GRAPH) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    OSD_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_step();
    hipError_t ce = hipStreamEndCapture(s, &graph);
    if (rc != OSD_OK) { if (graph) { hipError_t e = hipGraphDestroy(graph); (void)e; } return rc; }
    OSD_HIP(ce);
    OSD_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    sl.graph = graph;
    sl.exec = exec;
    for (int it = 0; it < T; ++it) OSD_HIP(hipGraphLaunch(exec, s));
  } else {
    for (int it = 0; it < T; ++it) OSD_TRY(enqueue_step());
  }
  return OSD_OK;
}

// y = x W^T + b on the bf16 matrix pipe (osd_op_linear under precision 1): both operands are split here, per call
int split_op_linear(osd_handle* h, const float* x, const float* w, const float* b, int64_t n, int K, int N, float* y) {
  hipStream_t s = h->stream;
  const int64_t ux = b3_units(n, K), uw = b3_units(N, K);
  OSD_TRY(ensure_arena(&h->main, (ux + uw) * 4 + 64));
  uint4* px = reinterpret_cast<uint4*>(h->main.arena);
  uint4* pw = px + ux;
  OSD_HIP(launch_pack(s, x, K, n, K, px));
  OSD_HIP(launch_pack(s, w, K, N, K, pw));
  const int nkb = b3_nkb(K);
  OSD_HIP(launch_b3<EpiB3Bias>(s, Bf3Args{pw, nkb, px, nkb, nullptr, 0, N, (int)n, nullptr}, EpiB3Bias::Args{b, y, N}));
  return OSD_OK;
}

int split_prepare(osd_handle* h, hipStream_t s) { return split_pack_weights(h, s); }

}  // namespace osd
