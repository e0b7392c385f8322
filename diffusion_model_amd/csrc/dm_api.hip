// dm_api.hip — C ABI (include/dm_sampler.h) and launch logic of libdm_sampler.so.
// Host code: geometry selection, packed-weight layout, workspace carving, the UNet launch
// sequence (diffusion/model.py:120-151), the sampling loop and its hipGraph capture
// (diffusion/sampler.py:45-76,136-172).  No torch types; plain pointers and sizes.
#include "dm_kernels.hip"
#include "dm_kernels_bf16.hip"
#include "dm_kernels_x6.hip"
#include "dm_kernels_x6p.hip"
#include "dm_kernels_x6c3.hip"
#include "dm_kernels_init5.hip"
#include "dm_kernels_generic.hip"
#include "dm_kernels_lowres.hip"
#include "../../include/dm_sampler.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Per-device facts, looked up once per device ordinal (function attributes and the CU count are
// per device: a process may run models on several GPUs).
constexpr int DM_MAX_DEVICES = 64;
constexpr int DM_MAX_BRANCHES = 8;  // dm_sample's parallel sub-batch chains (sample_split): the default rule asks for 4, DM_SPLIT for up to 8
inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= DM_MAX_DEVICES) d = 0;
  return d;
}
inline unsigned cu_count(int dev) {
  static int cus[DM_MAX_DEVICES] = {0};
  if (cus[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev] = v;
  }
  return (unsigned)cus[dev];
}

// ------------------------------------------------------------------ per-launch profiler
// dm_unet_profile brackets every launch of one forward with hipEvents (on the launch stream)
// and books its algorithmic FLOPs / bytes per kernel class; inactive otherwise.
struct ProfRec { int cls; hipEvent_t a, b; };
struct Prof {
  std::vector<ProfRec> recs;
  double flops[DM_K_COUNT] = {0};
  double bytes[DM_K_COUNT] = {0};
};
thread_local Prof* g_prof = nullptr;

struct ProfScope {
  int cls; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; bool on;
  ProfScope(int cls_, hipStream_t s_, double flops, double bytes) : cls(cls_), s(s_), on(g_prof != nullptr) {
    if (!on) return;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    g_prof->flops[cls] += flops; g_prof->bytes[cls] += bytes;
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    g_prof->recs.push_back({cls, a, b});
  }
};

// ------------------------------------------------------------------ conv geometry
struct ConvGeo {
  int MT, th_log, tw_log, ni, tiles_x, tiles_per_img, hth, htw, Hb, Wb, pout, OH, OW;
  float cnt_out;
};

// Tile shape depends on the layer's resolution only (never on B), so per-image results are
// independent of the batch size and of how a batch is sharded over GPUs.
ConvGeo conv_geo(int kind, int H, int W, int Cout) {
  ConvGeo g;
  const bool down128 = kind == 6;  // kind 6: the 4x4 stride-2 conv (kind 1) on 128-pixel tiles, 16-channel K chunks
  if (down128) kind = 1;
  g.Hb = (kind == 1) ? H / 2 : H;
  g.Wb = (kind == 1) ? W / 2 : W;
  g.OH = (kind == 1) ? H / 2 : (kind == 2) ? 2 * H : H;
  g.OW = (kind == 1) ? W / 2 : (kind == 2) ? 2 * W : W;
  const int npix = g.Hb * g.Wb;
  g.MT = (kind == 1) ? (down128 ? 128 : 64) : (kind == 5 || npix >= 256) ? 128 : 64;  // kind 5 (Winograd): always 32 2x2-pixel tiles
  int tw = g.Wb < ((kind == 1) ? 8 : 16) ? g.Wb : ((kind == 1) ? 8 : 16);
  int th = g.Hb < g.MT / tw ? g.Hb : g.MT / tw;
  g.ni = g.MT / (th * tw);
  g.th_log = ilog2(th);
  g.tw_log = ilog2(tw);
  g.tiles_x = g.Wb / tw;
  g.tiles_per_img = (g.Hb / th) * g.tiles_x;
  const int s = (kind == 1) ? 2 : 1;
  const int halo = (kind == 2) ? 1 : (kind == 3) ? 0 : 2;
  g.hth = th * s + halo;
  g.htw = tw * s + halo;
  g.pout = g.tiles_per_img * (kind == 2 ? 4 : 1);
  g.cnt_out = (float)(th * tw) * (float)(Cout / DM_GROUPS);
  return g;
}

#ifdef DM_STAMPS
unsigned long long* g_stamps = nullptr;  // diagnostic build: set through dm_debug_set_stamps
#endif

#ifndef DM_X6_DEFAULT
#define DM_X6_DEFAULT 487
#endif
struct ConvCall {
  int kind;
  const float* x1; int c1;
  const float* x2; int c2;
  const float* wp; const float* bias; int cout;
  const float* stats_in; int pin; float cnt_in;
  const float* gamma; const float* beta;
  const float* temb; int temb_stride; int temb_step_stride;
  const int* step_ptr; int step_off;
  float* y; float* stats_out;
  size_t stats_cap;  // floats available behind stats_out (0 = caller-sized, unchecked)
  int B, H, W;
  const float* res; int no_silu;
  int bf16;  // activations and packed weights are bf16 (conv_bf16_kernel)
  int f16op; // bf16 path, layers behind a GroupNorm: MFMA operands as IEEE half (1: (hi, lo) half weight pairs, 2: one half weight)
  int in_wrap;  // input images n >= in_wrap read image n - in_wrap (3x3 Winograd / bf16 3x3 kernel, one image per tile)
  const float* amax1; const float* amax2; int namax1, namax2;  // x6 == 3 on raw inputs: the producers' maxima (ConvArgs.amax1)
  float* amax_out;                                            // x6 kernels, tiles of one image: maxima of this launch's output
  int x6;    // fp32 activations, weights packed as three bf16 split terms (conv_x6_kernel); kinds 0, 1/6, 2; 2: two-term form (opt-in);
             // 3: two fp16 terms (kinds 1/6, 2 behind a GroupNorm, one image per tile; the pack is pack_conv's kind 13 / 14)
  const float* init_x; const float* init_w; const float* init_b; int init_cin;  // folded init_conv
};

template <int MB, int KIND, int KC, int GEO = 0>
int launch_conv_t(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_done[DM_MAX_DEVICES] = {false};  // per device; benign race: setting the attribute twice is harmless
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<MB, KIND, KC, GEO>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  constexpr bool UP = (KIND == 2 || KIND == 4);
  const int cls = KIND == 5 ? DM_K_WINO : KIND == 0 ? (MB == 2 ? DM_K_CONV3_128 : DM_K_CONV3_64)
                            : KIND == 1 ? DM_K_DOWN : KIND == 3 ? DM_K_ATTN : (MB == 2 ? DM_K_UP_128 : DM_K_UP_64);
  const double opix = (double)a.B * a.Hb * a.Wb * (UP ? 4 : 1);
  const double ipix = (double)a.B * a.H * a.W;
  // (KIND 5 books the ALGORITHMIC 3x3 count, 9 multiply-adds per tap set; it executes 4 per output)
  const double taps = (KIND == 0 || KIND == 5) ? 9 : KIND == 1 ? 16 : KIND == 3 ? 1 : 4;
  const int cin = a.C1 + a.C2;
  ProfScope ps(cls, s, 2.0 * opix * a.Cout * taps * cin, 4.0 * (ipix * cin + opix * a.Cout + taps * (UP ? 4 : 1) * cin * a.Cout));
  hipLaunchKernelGGL((conv_mfma_kernel<MB, KIND, KC, GEO>), grid, dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int MB, int KIND, bool INIT = false, int F16 = 0>
int launch_conv_bf16_t(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_done[DM_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MB, KIND, INIT, F16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  constexpr bool UP = (KIND == 2 || KIND == 4);
  const int cls = KIND == 0 ? (MB == 2 ? DM_K_CONV3_128 : DM_K_CONV3_64) : KIND == 1 ? DM_K_DOWN : (MB == 2 ? DM_K_UP_128 : DM_K_UP_64);
  const double opix = (double)a.B * a.Hb * a.Wb * (UP ? 4 : 1);
  const double ipix = (double)a.B * a.H * a.W;
  const double taps = KIND == 0 ? 9 : KIND == 1 ? 16 : 4;
  const int cin = a.C1 + a.C2;
  ProfScope ps(cls, s, 2.0 * opix * a.Cout * taps * cin, 2.0 * (ipix * cin + opix * a.Cout + taps * (UP ? 4 : 1) * cin * a.Cout));
  hipLaunchKernelGGL((conv_bf16_kernel<MB, KIND, INIT, F16>), grid, dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int MB, int KIND, bool SW, int NS = 3, bool F16 = false>
int launch_conv_x6_t(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_done[DM_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x6_kernel<MB, KIND, SW, NS, F16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  constexpr bool UP = (KIND == 4 || KIND == 2);
  const int cls = KIND == 0 ? DM_K_X6_CONV3 : KIND == 1 ? DM_K_X6_DOWN : DM_K_X6_UP;
  const double opix = (double)a.B * a.Hb * a.Wb * (UP ? 4 : 1);
  const double ipix = (double)a.B * a.H * a.W;
  const double taps = KIND == 0 ? 9 : KIND == 1 ? 16 : 4;
  const int cin = a.C1 + a.C2;
  ProfScope ps(cls, s, 2.0 * opix * a.Cout * taps * cin, 4.0 * (ipix * cin + opix * a.Cout) + 6.0 * taps * (UP ? 4 : 1) * cin * a.Cout);
  hipLaunchKernelGGL((conv_x6_kernel<MB, KIND, SW, NS, F16>), grid, dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int MB, int NS, bool TF, bool F16 = false>
int launch_conv_x6_down_t(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_done[DM_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x6_down_kernel<MB, NS, TF, F16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  const double opix = (double)a.B * a.Hb * a.Wb, ipix = (double)a.B * a.H * a.W;
  ProfScope ps(DM_K_X6_DOWN, s, 2.0 * opix * a.Cout * 16.0 * a.C1, 4.0 * (ipix * a.C1 + opix * a.Cout) + 6.0 * 16.0 * a.C1 * a.Cout);
  hipLaunchKernelGGL((conv_x6_down_kernel<MB, NS, TF, F16>), grid, dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

// diagnostics only (read once: getenv is neither cheap nor safe beside a concurrent setenv)
static bool dbg_noswap() {
  static const bool v = getenv("DM_DEBUG_NOSWAP") != nullptr;
  return v;
}

template <bool TF, int HL>
int launch_conv_x6_c3_t(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_done[DM_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x6_c3_kernel<TF, HL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  const double pix = (double)a.B * a.H * a.W;
  const int cin = a.C1 + a.C2;
  ProfScope ps(DM_K_X6_CONV3, s, 2.0 * pix * a.Cout * 9.0 * cin, 4.0 * pix * (cin + a.Cout) + 4.0 * 9.0 * cin * a.Cout);
  hipLaunchKernelGGL((conv_x6_c3_kernel<TF, HL>), grid, dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

// The fp32 convolutions on the bf16 matrix pipe (three-way split operands, conv_x6_kernel).  Same tile
// decomposition (hence the same GroupNorm partial counts) as the other arithmetic variants.
int launch_conv_x6(const ConvCall& c, hipStream_t s) {
  const int cin = c.c1 + c.c2;
  int kind = c.kind == 6 ? 1 : c.kind;
  if (kind != 0 && kind != 1 && kind != 2) return fail("conv (split-bf16): kind %d is not 3x3 / 4x4 stride 2 / transposed 4x4", c.kind);
  if (c.res || c.init_x || c.in_wrap) return fail("conv (split-bf16): no residual / folded init_conv / shared-input variant");
  if (c.c1 % 64 || c.c2 % 64 || c.cout % 64 || !is_pow2(c.c1) || (c.c2 && !is_pow2(c.c2)) || !is_pow2(c.cout))
    return fail("conv (split-bf16): channel counts must be powers of two >= 64 (got %d+%d -> %d)", c.c1, c.c2, c.cout);
  if (!is_pow2(c.H) || !is_pow2(c.W)) return fail("conv: spatial dims must be powers of two (got %dx%d)", c.H, c.W);
  if (kind == 1 && (c.H < 2 || c.W < 2)) return fail("conv: stride-2 conv needs H,W >= 2");
  if (c.stats_in && c.c2) return fail("conv: input transform and channel concat are exclusive");
  if (c.B <= 0) return fail("conv: empty batch");
  if (kind == 2 && c.H * c.W < 16) return fail("conv (split-bf16): transposed conv needs >= 16 input pixels");
  const bool fused_up = kind == 2 && cin == 64;  // one K chunk: all four parities from one tile; else one parity per workgroup (grid.z)
  ConvGeo g = conv_geo(c.kind, c.H, c.W, c.cout);
  if (c.stats_out && c.stats_cap && (size_t)c.B * g.pout * DM_GROUPS * 2 > c.stats_cap)
    return fail("conv: statistics buffer too small (%zu floats for %d images x %d partials)", c.stats_cap, c.B, g.pout);
  const int th = 1 << g.th_log, tw = 1 << g.tw_log;
  if (kind == 1) { g.hth = th + 1; g.htw = tw + 1; }        // one input phase at a time: (th+1) x (tw+1) pixels
  if (kind == 2) { g.hth = th + (fused_up ? 2 : 1); g.htw = tw + (fused_up ? 2 : 1); }
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.x1 = c.x1; a.x2 = c.x2; a.wp = c.wp; a.bias = c.bias; a.y = c.y;
  a.stats_in = c.stats_in; a.gamma = c.gamma; a.beta = c.beta; a.temb = c.temb;
  a.step_ptr = c.step_ptr; a.stats_out = c.stats_out; a.cnt_in = c.cnt_in; a.pin = c.pin;
  a.temb_stride = c.temb_stride; a.temb_step_stride = c.temb_step_stride; a.step_off = c.step_off;
  a.amax1 = c.amax1; a.amax2 = c.amax2; a.namax1 = c.namax1; a.namax2 = c.namax2; a.amax_out = c.amax_out;
  a.B = c.B; a.H = c.H; a.W = c.W; a.C1 = c.c1; a.C2 = c.c2; a.Cout = c.cout;
  a.Hb = g.Hb; a.Wb = g.Wb; a.th_log = g.th_log; a.tw_log = g.tw_log; a.ni = g.ni;
  a.tiles_x = g.tiles_x; a.tiles_per_img = g.tiles_per_img; a.hth = g.hth; a.htw = g.htw;
  a.mg_htw = (unsigned)((0x100000000ull + g.htw - 1) / g.htw);
  a.mg_hth = (unsigned)((0x100000000ull + g.hth - 1) / g.hth);
  a.pout = g.pout;
  a.tf_silu = c.no_silu ? 0 : 1;
  a.h_log = ilog2(c.H); a.w_log = ilog2(c.W); a.c1_log = ilog2(c.c1); a.c2_log = c.c2 ? ilog2(c.c2) : 0;
  a.cout_log = ilog2(c.cout);
  a.tpi_log = ilog2(g.tiles_per_img); a.tx_log = ilog2(g.tiles_x);
  a.g_log = ilog2(c.cout / DM_GROUPS);
  a.g1_log = ilog2(c.c1 / DM_GROUPS);
  (void)cin;
  const size_t np = (size_t)g.ni * g.hth * g.htw;
  if (np * 8 > 3 * 256 * 8) return fail("conv (split-bf16): tile of %zu halo pixels is too large", np);
  const size_t epi_f = (size_t)g.MT * 68 + 128;  // output staging tile + reduction words
  const int ns = c.x6 >= 2 ? 2 : 3;  // split terms per operand
  const bool f16 = c.x6 == 3;        // two fp16 terms (inputs behind a GroupNorm)
  if (f16 && ((!c.stats_in && !(kind == 0 && c.amax1 && (!c.c2 || c.amax2))) || g.ni != 1 || dbg_noswap()))
    return fail("conv (fp16 split): inputs behind a GroupNorm (or, 3x3: raw inputs with their producers' maxima), tiles of one image");
  if (c.amax_out && g.ni != 1) return fail("conv: output maxima need tiles of one image");
  size_t lds_f = (c.stats_in ? (size_t)(3 * g.ni) * c.c1 : 0) + (ns * np * 72 + 1) / 2 + 128;  // + reduction words behind the planes
  if (lds_f < epi_f) lds_f = epi_f;
  const size_t lds = lds_f * sizeof(float);
  if (lds > 160 * 1024) return fail("conv (split-bf16): tile needs %zu B of LDS (> 160 KiB)", lds);
  // stride-2 conv, one image per tile: the kernel with its staging pipelined into the K loop (dm_kernels_x6p.hip)
  static const bool x6_nopipe = getenv("DM_DEBUG_X6_NOPIPE") != nullptr;  // diagnostics only
  const bool pipe = kind == 1 && g.ni == 1 && g.tw_log == 3 && g.hth == g.MT / 8 + 1 && g.htw == 9 && !c.c2 && !(c.stats_in && c.no_silu) && !x6_nopipe && !dbg_noswap();
  if (pipe) {
    const size_t lds_p = ((c.stats_in ? (size_t)3 * c.c1 : 0) + 64) * sizeof(float) + (size_t)2 * ns * np * 40 * 2;
    if (lds_p > 80 * 1024) return fail("conv (split-bf16, stride 2): tile needs %zu B of LDS (> 80 KiB)", lds_p);
    dim3 gridp((unsigned)(c.B * g.tiles_per_img), (unsigned)(c.cout / 64), 1u);
    a.ntiles = (int)gridp.x;
    const unsigned slots = 2u * cu_count(current_device());
    const unsigned resident = slots / gridp.y ? slots / gridp.y : 1u;
    if (gridp.x > resident) gridp.x = resident;
    a.ngroups = c.B;
    a.xcd_map = (gridp.x % 8 == 0 && gridp.y == 1 && c.B >= 64) ? 1 : 0;
#ifdef DM_STAMPS
    a.stamps = g_stamps;
#endif
#define DM_X6P(MB_) (f16 ? launch_conv_x6_down_t<MB_, 2, true, true>(a, gridp, lds_p, s) \
                     : ns == 2 ? (c.stats_in ? launch_conv_x6_down_t<MB_, 2, true>(a, gridp, lds_p, s) : launch_conv_x6_down_t<MB_, 2, false>(a, gridp, lds_p, s)) \
                               : (c.stats_in ? launch_conv_x6_down_t<MB_, 3, true>(a, gridp, lds_p, s) : launch_conv_x6_down_t<MB_, 3, false>(a, gridp, lds_p, s)))
    return g.MT == 128 ? DM_X6P(2) : DM_X6P(1);
#undef DM_X6P
  }
  // 3x3, two fp16 terms, 8x16-pixel tiles of one image: the kernel with its staging pipelined into the K loop (dm_kernels_x6c3.hip)
  static const bool x6_c3 = getenv("DM_DEBUG_X6_NOC3") == nullptr;  // (diagnostics: conv_x6_kernel instead)
  if (x6_c3 && f16 && kind == 0 && g.MT == 128 && g.th_log == 3 && g.tw_log == 4 && c.c1 <= 256 && !x6_nopipe &&
      c.H == c.W && c.H >= 16 && c.H <= 128 && (!c.c2 || c.c2 == c.c1) &&
      (c.stats_in ? c.pin <= 8 : (c.namax1 <= 64 && c.namax2 <= 64))) {
    const size_t lds_c = ((c.stats_in ? (size_t)6 * c.c1 : 0) + 64) * sizeof(float) + (size_t)2 * 2 * 180 * 40 * 2;
    dim3 gridc((unsigned)(c.B * g.tiles_per_img), (unsigned)(c.cout / 64), 1u);
    a.ntiles = (int)gridc.x;
    const unsigned slots = 2u * cu_count(current_device());
    const unsigned resident = slots / gridc.y ? slots / gridc.y : 1u;
    if (gridc.x > resident) gridc.x = resident;
    a.ngroups = c.B;
    a.xcd_map = (gridc.x % 8 == 0 && gridc.y == 1 && c.B >= 64) ? 1 : 0;
#ifdef DM_STAMPS
    a.stamps = g_stamps;
#endif
#define DM_C3(HL_) (c.stats_in ? launch_conv_x6_c3_t<true, HL_>(a, gridc, lds_c, s) : launch_conv_x6_c3_t<false, HL_>(a, gridc, lds_c, s))
    return c.H == 16 ? DM_C3(4) : c.H == 32 ? DM_C3(5) : c.H == 64 ? DM_C3(6) : DM_C3(7);
#undef DM_C3
  }
  const int groups = (c.B + g.ni - 1) / g.ni;
  dim3 grid((unsigned)(groups * g.tiles_per_img), (unsigned)(c.cout / 64), (kind == 2 && !fused_up) ? 4u : 1u);
  a.ntiles = (int)grid.x;
  {
    static const unsigned slot_div = getenv("DM_DEBUG_SLOT_DIV") ? (unsigned)atoi(getenv("DM_DEBUG_SLOT_DIV")) : 1u;  // diagnostics only
    static const bool x6_2wg = getenv("DM_DEBUG_X6_2WG") != nullptr;  // diagnostics only
    const unsigned per_cu = lds > 80 * 1024 ? 1u : (lds <= 53 * 1024 && g.MT == 64 && !x6_2wg) ? 3u : 2u;  // (64-pixel tiles: <= 159 registers)
    static const unsigned x6_pc = getenv("DM_DEBUG_X6_PER_CU") ? (unsigned)atoi(getenv("DM_DEBUG_X6_PER_CU")) : 0u;  // diagnostics only
    const unsigned slots = (x6_pc && x6_pc < per_cu ? x6_pc : per_cu) * cu_count(current_device()) / (slot_div ? slot_div : 1u);
    const unsigned resident = slots / (grid.y * grid.z) ? slots / (grid.y * grid.z) : 1u;
    if (grid.x > resident) grid.x = resident;
    a.ngroups = groups;
    a.xcd_map = (grid.x % 8 == 0 && grid.y == 1 && grid.z == 1 && groups >= 64) ? 1 : 0;
  }
#ifdef DM_STAMPS
  a.stamps = g_stamps;
#endif
  // one image per tile: swapped MFMA operands, stores straight from the registers (x6_epilogue_sw)
  const bool x6_noswap = dbg_noswap();
  const bool sw = g.ni == 1 && !x6_noswap;
#define DM_X6H(MB_, K_) launch_conv_x6_t<MB_, K_, true, 2, true>(a, grid, lds, s)
  if (f16) {
    if (kind == 0) return g.MT == 128 ? DM_X6H(2, 0) : DM_X6H(1, 0);
    if (kind == 1) return g.MT == 128 ? DM_X6H(2, 1) : DM_X6H(1, 1);
    if (!fused_up) return g.MT == 128 ? DM_X6H(2, 2) : DM_X6H(1, 2);
    return g.MT == 128 ? DM_X6H(2, 4) : DM_X6H(1, 4);
  }
#undef DM_X6H
#define DM_X6(MB_, K_) (ns == 2 ? (sw ? launch_conv_x6_t<MB_, K_, true, 2>(a, grid, lds, s) : launch_conv_x6_t<MB_, K_, false, 2>(a, grid, lds, s)) \
                               : (sw ? launch_conv_x6_t<MB_, K_, true>(a, grid, lds, s) : launch_conv_x6_t<MB_, K_, false>(a, grid, lds, s)))
  if (kind == 0) return g.MT == 128 ? DM_X6(2, 0) : DM_X6(1, 0);
  if (kind == 1) return g.MT == 128 ? DM_X6(2, 1) : DM_X6(1, 1);
  if (!fused_up) return g.MT == 128 ? DM_X6(2, 2) : DM_X6(1, 2);
  return g.MT == 128 ? DM_X6(2, 4) : DM_X6(1, 4);
#undef DM_X6
}

int launch_conv(const ConvCall& c, hipStream_t s) {
  if (c.x6) return launch_conv_x6(c, s);
  const int cin = c.c1 + c.c2;
  if (c.kind < 0 || (c.kind > 3 && c.kind != 5 && c.kind != 6)) return fail("conv: unknown kind %d", c.kind);
  if (c.kind == 6 && (c.bf16 || (c.H / 2) * (c.W / 2) < 128)) return fail("conv: the 128-pixel-tile stride-2 variant is fp32, for outputs of >= 128 pixels");
  if (c.kind == 5 && (c.bf16 || c.res || c.H < 8 || c.W < 8))
    return fail("conv: the Winograd 3x3 path is fp32, without residual, for H,W >= 8 (got %dx%d)", c.H, c.W);
  if (c.c1 % 64 || c.c2 % 64 || c.cout % 64) return fail("conv: channel counts must be multiples of 64 (got %d+%d -> %d)", c.c1, c.c2, c.cout);
  if (c.cout != 64 && c.cout != 128 && c.cout != 256 && c.cout != 512 && c.stats_out)
    return fail("conv: GroupNorm(8) statistics need Cout in {64,128,256,512} (got %d)", c.cout);
  if (!is_pow2(c.H) || !is_pow2(c.W)) return fail("conv: spatial dims must be powers of two (got %dx%d)", c.H, c.W);
  if ((c.kind == 1 || c.kind == 6) && (c.H < 2 || c.W < 2)) return fail("conv: stride-2 conv needs H,W >= 2");
  if (c.stats_in && c.c2) return fail("conv: input transform and channel concat are exclusive");
  if (c.B <= 0) return fail("conv: empty batch");
  ConvGeo g = conv_geo(c.kind, c.H, c.W, c.cout);
  if (c.stats_out && c.stats_cap && (size_t)c.B * g.pout * DM_GROUPS * 2 > c.stats_cap)
    return fail("conv: statistics buffer too small (%zu floats for %d images x %d partials)", c.stats_cap, c.B, g.pout);
  // transposed conv with a single K chunk: all four output parities from one staged tile
  if (c.bf16 && (c.kind == 3 || c.res)) return fail("conv: the bf16 path has no 1x1 / residual variant");
  // (not below 4x4 input pixels: the tiny-image statistics path needs the staged output tile)
  const bool fused_up = c.kind == 2 && cin == 64 && g.Hb * g.Wb >= 16;
  if (fused_up) { g.hth += 1; g.htw += 1; }
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.x1 = c.x1; a.x2 = c.x2; a.wp = c.wp; a.bias = c.bias; a.y = c.y;
  a.stats_in = c.stats_in; a.gamma = c.gamma; a.beta = c.beta; a.temb = c.temb;
  a.step_ptr = c.step_ptr; a.stats_out = c.stats_out; a.cnt_in = c.cnt_in; a.pin = c.pin;
  a.temb_stride = c.temb_stride; a.temb_step_stride = c.temb_step_stride; a.step_off = c.step_off;
  a.B = c.B; a.H = c.H; a.W = c.W; a.C1 = c.c1; a.C2 = c.c2; a.Cout = c.cout;
  a.Hb = g.Hb; a.Wb = g.Wb; a.th_log = g.th_log; a.tw_log = g.tw_log; a.ni = g.ni;
  a.tiles_x = g.tiles_x; a.tiles_per_img = g.tiles_per_img; a.hth = g.hth; a.htw = g.htw;
  a.mg_htw = (unsigned)((0x100000000ull + g.htw - 1) / g.htw);
  a.mg_hth = (unsigned)((0x100000000ull + g.hth - 1) / g.hth);
  a.pout = g.pout;
  a.res = c.res; a.tf_silu = c.no_silu ? 0 : 1;
  if (!is_pow2(c.c1) || (c.c2 && !is_pow2(c.c2)))
    return fail("conv: input channel counts must be powers of two (got %d+%d)", c.c1, c.c2);
  a.h_log = ilog2(c.H); a.w_log = ilog2(c.W); a.c1_log = ilog2(c.c1); a.c2_log = c.c2 ? ilog2(c.c2) : 0;
  a.cout_log = is_pow2(c.cout) ? ilog2(c.cout) : -1;
  a.tpi_log = ilog2(g.tiles_per_img); a.tx_log = ilog2(g.tiles_x);
  a.g_log = is_pow2(c.cout) ? ilog2(c.cout / DM_GROUPS) : 3;  // (only read when statistics are requested: Cout in {64..512})
  a.g1_log = ilog2(c.c1 / DM_GROUPS);
  a.init_x = c.init_x; a.init_w = c.init_w; a.init_b = c.init_b; a.init_cin = c.init_cin;
  if (c.init_x && ((c.kind != 0 && c.kind != 5) || c.stats_in || c.c2)) return fail("conv: folded init_conv needs a plain 3x3 conv");
  if (c.in_wrap) {
    if (g.ni != 1 || !(c.kind == 5 || (c.kind == 0 && c.bf16)) || c.c2 || c.init_x) return fail("conv: shared-input batches need a 3x3 layer with one image per tile");
    a.in_wrap = c.in_wrap;
  }
  const size_t init_f = c.init_x ? (size_t)g.ni * c.init_cin * (g.hth + 2) * (g.htw + 2) + 4 + (size_t)c.init_cin * 9 * 64 + 64 + 8 : 0;
#ifdef DM_STAMPS
  a.stamps = g_stamps;
#endif
  const size_t np = (size_t)g.ni * g.hth * g.htw;
  const size_t epi_f = (size_t)g.MT * 68 + 128;  // output staging tile + reduction words
  auto lds_floats = [&](int kc_) {
    const size_t f = np * (kc_ + 4) + (c.stats_in ? (size_t)(3 * g.ni) * c.c1 : 0) + (fused_up ? 128 : 0) + init_f;
    return f < epi_f ? epi_f : f;
  };
  int kc = (c.kind == 6) ? 16 : (c.kind == 1) ? 32 : 64;  // K chunk (input channels staged at a time)
  // Images below 4x4 pixels ride 16-64 to a tile, each with its own halo: halve the K chunk when the halo tile
  // would not fit the CU's LDS (the deepest levels of 8x8 / 16x16 models)
  if (!c.bf16 && !c.init_x && g.MT == 64 && (c.kind == 0 || c.kind == 1) && lds_floats(kc) * sizeof(float) > 160 * 1024) kc /= 2;
  {
    const int ppp = c.bf16 ? 32 : 256 / (kc / 4);  // 256 threads / (16-byte loads per pixel per chunk)
    a.ppp_q = ppp / g.htw; a.ppp_r = ppp % g.htw;
  }
  size_t lds_f = lds_floats(kc);
  if (c.kind == 5) {
    if (lds_f < DM_WINO_EPI_FLOATS) lds_f = DM_WINO_EPI_FLOATS;
    lds_f += 208;  // staging descriptor table
    if (np > 208) return fail("conv: Winograd tile has %zu halo pixels (> 208)", np);
    if ((np + 1) * 68 + (c.stats_in ? (size_t)(3 * g.ni) * c.c1 : 0) + 208 > lds_f) return fail("conv: no room for the staging dump row");
  }
  if (c.bf16) {
    lds_f = (np * (64 + 8) + 1) / 2 + (c.stats_in ? (size_t)(3 * g.ni + 1) * c.c1 : 0) + init_f;
    const size_t e2 = (size_t)g.MT * 36 + 128;
    if (fused_up) lds_f += e2 + 8;  // the four parities' outputs are staged beside the input tile, not over it
    else if (lds_f < e2) lds_f = e2;
  }
  const size_t lds = lds_f * sizeof(float);
  if (lds > 160 * 1024) return fail("conv: tile needs %zu B of LDS (> 160 KiB)", lds);
  const int groups = (c.B + g.ni - 1) / g.ni;
  dim3 grid((unsigned)(groups * g.tiles_per_img), (unsigned)(c.cout / 64), (c.kind == 2 && !fused_up) ? 4u : 1u);
  a.ntiles = (int)grid.x;
  {  // persistent kernels: at most 3 (Winograd: 2) resident workgroups per CU
    static const unsigned slot_div = getenv("DM_DEBUG_SLOT_DIV") ? (unsigned)atoi(getenv("DM_DEBUG_SLOT_DIV")) : 1u;  // diagnostics only
    const unsigned slots = (c.kind == 5 ? 2u : 3u) * cu_count(current_device()) / (slot_div ? slot_div : 1u);  // (measured: 2 vs 3 per CU for the direct kernels is a wash)
    const unsigned resident = slots / (grid.y * grid.z) ? slots / (grid.y * grid.z) : 1u;
    if (grid.x > resident) grid.x = resident;
    a.ngroups = groups;
    // XCD-aware tile order once every XCD has several rounds of work (exact only for 8 | grid.x)
    a.xcd_map = (grid.x % 8 == 0 && grid.y == 1 && grid.z == 1 && groups >= 64) ? 1 : 0;
  }
  if (c.f16op && (!c.bf16 || !c.stats_in || c.init_x || c.c2 || c.f16op > 2)) return fail("conv: half operands are for the bf16 path's layers behind a GroupNorm");
  if (c.bf16 && c.f16op) {
#define DM_BFH(MB_, K_) (c.f16op == 2 ? launch_conv_bf16_t<MB_, K_, false, 2>(a, grid, lds, s) : launch_conv_bf16_t<MB_, K_, false, 1>(a, grid, lds, s))
    if (c.kind == 0) return g.MT == 128 ? DM_BFH(2, 0) : DM_BFH(1, 0);
    if (fused_up) return g.MT == 128 ? DM_BFH(2, 4) : DM_BFH(1, 4);
    if (c.kind == 2) return g.MT == 128 ? DM_BFH(2, 2) : DM_BFH(1, 2);
    return DM_BFH(1, 1);
#undef DM_BFH
  }
  if (c.bf16) {
    if (c.kind == 0 && c.init_x) return g.MT == 128 ? launch_conv_bf16_t<2, 0, true>(a, grid, lds, s) : launch_conv_bf16_t<1, 0, true>(a, grid, lds, s);
    if (c.kind == 0) return g.MT == 128 ? launch_conv_bf16_t<2, 0>(a, grid, lds, s) : launch_conv_bf16_t<1, 0>(a, grid, lds, s);
    if (fused_up) return g.MT == 128 ? launch_conv_bf16_t<2, 4>(a, grid, lds, s) : launch_conv_bf16_t<1, 4>(a, grid, lds, s);
    if (c.kind == 2) return g.MT == 128 ? launch_conv_bf16_t<2, 2>(a, grid, lds, s) : launch_conv_bf16_t<1, 2>(a, grid, lds, s);
    return launch_conv_bf16_t<1, 1>(a, grid, lds, s);
  }
  if (c.kind == 5) {
    const bool geo1 = g.th_log == 3 && g.tw_log == 4 && g.ni == 1 && c.c1 == 64 && (c.c2 == 0 || c.c2 == 64) && c.cout == 64;  // GEO=1
    const bool geo2 = c.H == 8 && c.W == 8 && g.ni == 2 && c.c1 == 64 && (c.c2 == 0 || c.c2 == 64) && c.cout == 64;  // GEO=2
    const bool geo3 = g.th_log == 3 && g.tw_log == 4 && g.ni == 1 && c.c1 == 256 && (c.c2 == 0 || c.c2 == 256) && c.cout == 256;  // GEO=3
    return geo1 ? launch_conv_t<2, 5, 64, 1>(a, grid, lds, s) : geo2 ? launch_conv_t<2, 5, 64, 2>(a, grid, lds, s)
         : geo3 ? launch_conv_t<2, 5, 64, 3>(a, grid, lds, s) : launch_conv_t<2, 5, 64>(a, grid, lds, s);
  }
  const bool c64 = c.c1 == 64 && (c.c2 == 0 || c.c2 == 64) && c.cout == 64 && g.ni == 1;  // compile-time geometry (GEO=1)
  if (c.kind == 6) return (c64 && g.th_log == 4 && g.tw_log == 3) ? launch_conv_t<2, 1, 16, 1>(a, grid, lds, s) : launch_conv_t<2, 1, 16>(a, grid, lds, s);
  if (fused_up && g.MT == 128 && c64 && c.c2 == 0 && g.th_log == 3 && g.tw_log == 4) return launch_conv_t<2, 4, 64, 1>(a, grid, lds, s);
  if (fused_up) return g.MT == 128 ? launch_conv_t<2, 4, 64>(a, grid, lds, s) : launch_conv_t<1, 4, 64>(a, grid, lds, s);
  (void)cin;
  if (c.kind == 0 && kc == 32) return launch_conv_t<1, 0, 32>(a, grid, lds, s);
  if (c.kind == 1 && kc == 16) return launch_conv_t<1, 1, 16>(a, grid, lds, s);
  if (c.kind == 0) return g.MT == 128 ? launch_conv_t<2, 0, 64>(a, grid, lds, s) : launch_conv_t<1, 0, 64>(a, grid, lds, s);
  if (c.kind == 2) return g.MT == 128 ? launch_conv_t<2, 2, 64>(a, grid, lds, s) : launch_conv_t<1, 2, 64>(a, grid, lds, s);
  if (c.kind == 3) return g.MT == 128 ? launch_conv_t<2, 3, 64>(a, grid, lds, s) : launch_conv_t<1, 3, 64>(a, grid, lds, s);
  return launch_conv_t<1, 1, 32>(a, grid, lds, s);
}

size_t conv_packed_floats(int kind, int cin, int cout) {
  if (kind >= 13 && kind <= 15) return (size_t)(kind == 15 ? 9 : 16) * cin * cout + 4;  // two fp16 planes + the float that undoes the weight scale
  if (kind >= 7 && kind <= 9) {  // split-bf16 form of kinds 0, 1, 2: three bf16 terms per weight
    const int taps = kind == 7 ? 9 : 16;
    return ((size_t)taps * cin * cout * 3 + 1) / 2;
  }
  if (kind == 6) kind = 1;  // same packed weights
  if (kind >= 10 && kind <= 12) kind -= 10;  // any-geometry kernels: the direct form, output channels padded to 32
  const int taps = kind == 0 ? 9 : kind == 3 ? 1 : 16;  // KIND 2: 4 parities x 4 taps; KIND 5: 16 transform points
  return (size_t)taps * cin * ((cout + 31) / 32 * 32);
}

int pack_conv(int kind, const float* w, int cin, int cout, float* out, hipStream_t s, int bf16 = 0) {
  if (kind == 6) kind = 1;
  if (kind >= 10 && kind <= 12) kind -= 10;
  if (!bf16 && kind <= 2 && (cin % 16 || cout % 32)) {  // any-geometry form (gconv_kernel): cin % 8, any cout (zero-padded to 32)
    if (cin % 8 || cin <= 0 || cout <= 0) return fail("pack_conv: input channels must be a positive multiple of 8 (got %d -> %d)", cin, cout);
    const size_t total = conv_packed_floats(kind, cin, cout);
    hipLaunchKernelGGL(gpack_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kind, w, cin, cout, out, total);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (cin % 16 || cout % 32) return fail("pack_conv: cin %% 16 / cout %% 32");
  if (kind >= 13 && kind <= 15) {  // two-way fp16 split of kinds 1 / 2 / 0 (dm_kernels_x6.hip)
    const size_t total = (size_t)(kind == 15 ? 9 : 16) * cin * cout;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1024), 0, s, w, total, out + total);
    hipLaunchKernelGGL(pack_conv_weight_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kind == 15 ? 0 : kind - 12, w, cin, cout,
                       reinterpret_cast<bf16_t*>(out), total, out + total);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (kind >= 7 && kind <= 9) {
    const size_t total = (size_t)(kind == 7 ? 9 : 16) * cin * cout;
    hipLaunchKernelGGL(pack_conv_weight_x6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kind - 7, w, cin, cout,
                       reinterpret_cast<bf16_t*>(out), total);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const size_t total = conv_packed_floats(kind, cin, cout);
  if (bf16) {
    if (kind == 3 || kind == 5) return fail("pack_conv: no bf16 1x1 / Winograd variant");
    hipLaunchKernelGGL(pack_conv_weight_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kind, w, cin, cout,
                       reinterpret_cast<bf16_t*>(out), total, bf16 == 2 ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (kind == 5) hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, cin, cout, out, total);
  else hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kind, w, cin, cout, out, total);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// ------------------------------------------------------------------ model container
struct BlockOff { size_t tmw, tmb, c1, c1b, c2, c2b, tr, trb, g1, b1, g2, b2, c1w, c2w, c1x, c2x, trx, trh, c2h, c1h; };  // c1w/c2w: Winograd form; *x: split-bf16 form; *h: two fp16 terms

// A captured chunk of S sampling steps.  The nodes hold workspace addresses only (dm_sample stages
// x, timesteps, coefficients, class embeddings and noise into the workspace), so the key is the
// workspace, the weights and everything the carve layout / launch sequence depends on.
struct GraphEntry {
  const void *packed, *ws;
  int B, H, W, S, n, cfg, noise, bf16, nsplit;
  float guidance;
  hipGraph_t graph;
  hipGraphExec_t exec;
};

struct AttnOff { size_t ng, nb, qkv, qkvb, proj, projb; };

struct dm_unet {
  int Cin, Cout, C, E;
  int bf16;            // 1: bf16 activations + conv weights (fp32 accumulate / statistics)
  int bf_h;            // bf16 path: MFMA operands of the layers behind a GroupNorm in down1 / down2.conv2 / up3 as IEEE half
                       // (0: bf16 as everywhere else, 1: half with (hi, lo) weight pairs, 2: half with one weight; DM_BF16_F16)
  int x6;              // fp32 path: which convolutions run on the bf16 pipe with three-way split operands (conv_x6_kernel):
                       // bit 0 stride-2 4x4, bit 1 transposed 4x4, bit 2 3x3 below 8x8, bit 3 every 3x3
  int attn;            // build-defined attention blocks at 1/4 resolution (after down2 and up1)
  AttnOff att[2];
  size_t o_freqs, o_tm1w, o_tm1b, o_tm2w, o_tm2b, o_initw, o_initb;
  BlockOff blk[6];
  size_t o_b0, o_b0b, o_bg, o_bb, o_b3, o_b3b, o_fg, o_fb, o_fw, o_fwb, o_b0w, o_b3w, o_b0x, o_b3x, o_b0h, o_b3h;
  size_t o_i5w;        // init_conv composed with down1.conv1: nine weight sets (init5_kernel)
  int init5;           // 1: that composition replaces the two layers where the tile geometry allows (env DM_INIT5, default 1)
  int lowres;          // 1: the fused low-resolution kernel where it applies (env DM_LOWRES, default 1; read when the handle is created)
  int fast;            // C in {64,128,256,512}, in/out channels <= 4: the tuned kernels apply (to power-of-two image sides);
                       // otherwise every geometry runs on the any-geometry kernels (dm_kernels_generic.hip)
  size_t total;
  std::vector<GraphEntry> graphs;
  // dm_sample's parallel branches (sample_split): side streams and fork / join events, created on first use, per device
  hipStream_t split_streams[DM_MAX_DEVICES][DM_MAX_BRANCHES - 1] = {};
  hipEvent_t split_events[DM_MAX_DEVICES][DM_MAX_BRANCHES] = {};
};

namespace {

// bf16 path: does layer `which` (1: the stride-2 / transposed transform, 2: conv2 - both read SiLU(GroupNorm(.))) of block k
// take its MFMA operands as IEEE half (conv_bf16_kernel's F16 forms)?  The blocks at the highest resolution (down1, up3) and
// down2.conv2; a property of the model only (the fused low-resolution kernel keeps bf16 operands for every layer it covers, from
// down2.transform to up2.transform, and reads those layers' (hi, lo) bf16 packs).
inline bool bf_half_layer(const dm_unet* m, int k, int which) {
  return m->bf16 && m->bf_h && which != 0 && (k == 0 || k == 5 || (k == 1 && which == 2));
}

struct Workspace {
  float *A0, *T1, *T2, *D1, *D2, *D3, *BT0, *BT, *U1, *U2, *SA, *SB, *heads, *temb, *eps, *xx;
  float *xcur, *coefs, *cond, *noise;  // dm_sample's staged operands (graph nodes only see workspace addresses)
  float* AM[4];         // per-wave maxima of |.| of D1, D2, U1, U2, [B][amax_cap] (ConvArgs.amax1): scales of the two-fp16-term conv1
  size_t amax_cap;
  int64_t* ts;
  int* step;
  size_t stats_floats;  // capacity of SA and of SB
  size_t bytes;
};

// Largest number of GroupNorm partials per image that any layer of the launch sequence writes
// (conv_geo().pout over the 3x3, stride-2 and transposed convs at the four resolutions, both
// arithmetic variants): the statistics buffers are sized for it.
size_t max_stats_partials(int H, int W, int C) {
  int pm = 1;
  auto upd = [&](int kind, int h, int w) { const int q = conv_geo(kind, h, w, C).pout; if (q > pm) pm = q; };
  for (int l = 0; l < 4; ++l) {
    const int h = H >> l, w = W >> l;
    if (h < 1 || w < 1) break;
    upd(0, h, w);
    if (h >= 8 && w >= 8) upd(5, h, w);
    if (l < 3 && h >= 2 && w >= 2) {
      upd(1, h, w);
      if ((h / 2) * (w / 2) >= 128) upd(6, h, w);
    }
    if (l >= 1) upd(2, h, w);
  }
  return (size_t)pm;
}

// Do the tuned kernels (power-of-two tiles, Winograd / split-bf16 weight forms) cover HxW images of this model?
// Everything else the reference accepts (H, W divisible by 8; C divisible by 8) runs on dm_kernels_generic.hip.
bool tuned_geometry(const dm_unet* m, int H, int W) { return m->fast && is_pow2(H) && is_pow2(W) && H >= 8 && W >= 8; }

// Does a block's conv2 at h x w (its input is behind a GroupNorm) run as a direct 3x3 convolution with the operands as two fp16
// terms (conv_x6_kernel<.., 0, true, 2, true>)?  Where DM_X6 bit 6 asks for it and a tile holds one image; by geometry only.
bool conv2_f16(const dm_unet* m, int h, int ww, bool shared_input) {
  return !m->bf16 && (m->x6 & 64) && !(m->x6 & 16) && m->fast && !shared_input && h * ww >= 64 && is_pow2(h) && is_pow2(ww);
}

// ... and a block's conv1 (raw inputs: its producers' maxima give the power-of-two scale), DM_X6 bit 8?  (Not down1.conv1: that
// one is part of init5_kernel.)
bool conv1_f16(const dm_unet* m, int h, int ww) {
  return !m->bf16 && (m->x6 & 256) && !(m->x6 & 16) && m->fast && h * ww >= 64 && is_pow2(h) && is_pow2(ww);
}

// Is the sampler update folded into final_conv's store (FinalStep)?  fp32, C = 64, tuned geometry, no guidance (the caller's part).
bool fuses_step(const dm_unet* m, int H, int W) {
  static const bool off = getenv("DM_DEBUG_NO_STEP_FUSION") != nullptr;  // diagnostics only
  return !off && tuned_geometry(m, H, W) && !m->bf16 && m->C == 64 && m->Cin == m->Cout;
}

// Carve the workspace for a denoiser batch B (all offsets 256-B aligned).  Everything whose
// address a captured graph bakes in and whose size does not depend on the number of steps
// comes first, so those addresses are a function of (B, H, W) only; the step-count-dependent
// tables follow; `noise_floats` > 0 appends dm_sample's per-chunk noise staging area.
Workspace carve(const dm_unet* m, void* base, int B, int H, int W, int nsteps, size_t noise_floats = 0) {
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += align_up(floats * sizeof(float), 256);
    return p;
  };
  const size_t full = (size_t)B * H * W * m->C;
  w.step = reinterpret_cast<int*>(take(64));
  w.eps = take((size_t)B * m->Cout * H * W);
  w.xx = take((size_t)B * m->Cin * H * W);
  w.xcur = take((size_t)B * m->Cin * H * W);
  w.cond = take((size_t)B * m->E);
  w.A0 = take(full);  // init_conv output; reused for up3's output (u3)
  w.T1 = take(full);
  w.T2 = take(full);
  w.D1 = take(full / 4);
  w.D2 = take(full / 16);
  w.D3 = take(full / 64);
  w.BT0 = take(full / 64);
  w.BT = take(full / 64);
  w.U1 = take(full / 16);
  w.U2 = take(full / 4);
  w.stats_floats = (size_t)B * (tuned_geometry(m, H, W) ? max_stats_partials(H, W, m->C) : 1) * DM_GROUPS * 2;
  w.SA = take(w.stats_floats);
  w.SB = take(w.stats_floats);
  w.amax_cap = (tuned_geometry(m, H, W) && !m->bf16) ? (size_t)4 * ((m->C + 63) / 64) * max_stats_partials(H, W, m->C) : 0;
  for (int i = 0; i < 4; ++i) w.AM[i] = w.amax_cap ? take((size_t)B * w.amax_cap) : nullptr;
  const size_t nt = (size_t)(nsteps > B ? nsteps : B);
  w.heads = take(nt * 6 * m->C);
  w.temb = take(nt * m->E);
  w.coefs = take((size_t)nsteps * 8);
  w.ts = reinterpret_cast<int64_t*>(take((size_t)nsteps * 2));
  w.noise = noise_floats > 0 ? take(noise_floats) : nullptr;
  w.bytes = off;
  return w;
}

int launch_final(const float* u, int bf16, const float* stats, int pin, float cnt, const float* gamma, const float* beta,
                 const float* wt, const float* bias, float* out, int B, int H, int W, int C, int Cout, hipStream_t s,
                 const FinalStep* fs = nullptr);

int check_geometry(const dm_unet* m, int B, int H, int W) {
  if (B <= 0) return fail("batch must be positive (got %d)", B);
  // model.py:138-148: three 2x down-samplings and the skip concatenations need H, W divisible by 8 (the
  // reference raises a RuntimeError from torch.cat otherwise)
  if (H < 8 || W < 8 || H % 8 || W % 8)
    return fail("image height/width must be multiples of 8 (three 2x down-samplings; got %dx%d)", H, W);
  if (m->bf16 && (!tuned_geometry(m, H, W) || H < 16 || W < 16))
    return fail("the bf16 kernels need power-of-two image height/width >= 16 (got %dx%d); use fp32 for other sizes", H, W);
  if (m->attn && !tuned_geometry(m, H, W))
    return fail("the (build-defined) attention blocks need power-of-two image height/width (got %dx%d)", H, W);
  return 0;
}

int launch_time_embed(const dm_unet* m, const float* P, const int64_t* t, int nt, const float* cond,
                      float* temb, float* heads, hipStream_t s, const int* step_ptr = nullptr, int step_off = 0,
                      int t_stride = 1, const float* temb_in = nullptr) {
  TimeArgs a;
  a.step_ptr = step_ptr; a.step_off = step_off; a.t_stride = t_stride; a.temb_in = temb_in;
  a.freqs = P + m->o_freqs;
  a.w1 = P + m->o_tm1w; a.b1 = P + m->o_tm1b; a.w2 = P + m->o_tm2w; a.b2 = P + m->o_tm2b;
  for (int k = 0; k < 6; ++k) { a.hw[k] = P + m->blk[k].tmw; a.hb[k] = P + m->blk[k].tmb; }
  a.t = t; a.cond = cond; a.temb = temb; a.heads = heads; a.E = m->E; a.C = m->C;
  ProfScope ps(DM_K_TIME, s, 2.0 * nt * (2.0 * m->E * m->E + 6.0 * m->C * m->E), 4.0 * (2.0 * m->E * m->E + 6.0 * m->C * m->E));
  const size_t lds = (size_t)3 * DM_TIME_ROWS * m->E * sizeof(float);
  if (lds > 48 * 1024) {  // time_emb_dim > 512
    static bool attr_done[DM_MAX_DEVICES] = {false};
    const int dev = current_device();
    if (!attr_done[dev]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&time_embed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL(time_embed_kernel, dim3((unsigned)((nt + DM_TIME_ROWS - 1) / DM_TIME_ROWS)), dim3(256), lds, s, a, nt);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_bgemm(bool bkn, const GemmArgs& g, int batch, hipStream_t s) {
  if (g.M % 64 || g.N % 64 || g.K % 32) return fail("bgemm: M,N %% 64 and K %% 32 required (got %d,%d,%d)", g.M, g.N, g.K);
  ProfScope ps(DM_K_ATTN, s, 2.0 * batch * g.M * (double)g.N * g.K, 4.0 * batch * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
  dim3 grid(g.N / 64, g.M / 64, batch);
  if (bkn) hipLaunchKernelGGL(bgemm_mfma_kernel<true>, grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL(bgemm_mfma_kernel<false>, grid, dim3(256), 0, s, g);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int NB, int CB>
int launch_attention_t(const float* qkv, float* o, int B, int C, float scale, size_t lds, hipStream_t s) {
  static bool attr_done[DM_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_fused_kernel<NB, CB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((attention_fused_kernel<NB, CB>), dim3(NB, (unsigned)B), dim3(256), lds, s, qkv, o, C, scale);
  HIP_TRY(hipGetLastError());
  return 0;
}

// softmax(Q K^T / sqrt(C)) V for T in {64, 128, 256} tokens, C in {64, 128, 256, 512}: one fused kernel per
// (image, 64-query tile), scores in LDS.  Returns -1 when the shape is outside that set (caller falls back to
// the three-kernel form for longer sequences).
int launch_attention_fused(const float* qkv, float* o, int B, int T, int C, hipStream_t s) {
  const int nb = T / 64, cb = C / 64;
  if (T % 64 || C % 64 || (nb != 1 && nb != 2 && nb != 4) || (cb != 1 && cb != 2 && cb != 4 && cb != 8)) return -1;
  const size_t lds = ((size_t)64 * (T + 4) + 64 * 36 + (size_t)(T > C ? T : C) * 36) * sizeof(float);
  if (lds > 160 * 1024) return -1;
  const float scale = 1.0f / sqrtf((float)C);
  ProfScope ps(DM_K_ATTN, s, 4.0 * B * (double)T * T * C, 4.0 * B * (double)T * 4 * C);
#define DM_ATT(NB_, CB_) if (nb == NB_ && cb == CB_) return launch_attention_t<NB_, CB_>(qkv, o, B, C, scale, lds, s);
  DM_ATT(1, 1) DM_ATT(1, 2) DM_ATT(1, 4) DM_ATT(1, 8)
  DM_ATT(2, 1) DM_ATT(2, 2) DM_ATT(2, 4) DM_ATT(2, 8)
  DM_ATT(4, 1) DM_ATT(4, 2) DM_ATT(4, 4) DM_ATT(4, 8)
#undef DM_ATT
  return -1;
}

// Build-defined attention block on X [B,h,w,C] in place:
//   X += proj( softmax(Q K^T / sqrt(C)) V ),  [Q,K,V] = qkv_1x1( GroupNorm8(X) )
// GroupNorm statistics of X come as partials from the conv that produced it.
int attention_block(const dm_unet* m, const float* P, int which, float* X, const float* stats, int pin, float cnt,
                    const Workspace& w, int B, int h, int ww, int Hfull, int Wfull, hipStream_t s) {
  const int C = m->C, T = h * ww;
  const AttnOff& o = m->att[which];
  const size_t full = (size_t)B * Hfull * Wfull * C;
  if ((size_t)B * T * 4 * C > full || (size_t)B * T * T > full) return fail("attention: workspace too small for %d tokens", T);
  if (T % 64) return fail("attention: token count %d must be a multiple of 64", T);
  float* QKV = w.T1;
  float* O = w.T1 + (size_t)B * T * 3 * C;
  float* S = w.T2;
  ConvCall c;
  memset(&c, 0, sizeof c);
  c.kind = 3; c.x1 = X; c.c1 = C; c.wp = P + o.qkv; c.bias = P + o.qkvb; c.cout = 3 * C; c.y = QKV;
  c.stats_in = stats; c.pin = pin; c.cnt_in = cnt; c.gamma = P + o.ng; c.beta = P + o.nb; c.no_silu = 1;
  c.B = B; c.H = h; c.W = ww;
  if (launch_conv(c, s)) return 1;
  const int frc = launch_attention_fused(QKV, O, B, T, C, s);
  if (frc > 0) return 1;
  if (frc < 0) {  // sequences beyond 256 tokens: QK^T, row softmax and PV as three kernels, scores through HBM
  GemmArgs g;
  g.A = QKV; g.Bm = QKV + C; g.C = S; g.M = T; g.N = T; g.K = C; g.lda = 3 * C; g.ldb = 3 * C; g.ldc = T;
  g.sA = (size_t)T * 3 * C; g.sB = g.sA; g.sC = (size_t)T * T; g.alpha = 1.0f / sqrtf((float)C);
  if (launch_bgemm(false, g, B, s)) return 1;
  {
    const size_t rows = (size_t)B * T;
    ProfScope ps(DM_K_ATTN, s, 5.0 * rows * T, 8.0 * rows * T);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, S, rows, T);
    HIP_TRY(hipGetLastError());
  }
  g.A = S; g.Bm = QKV + 2 * C; g.C = O; g.M = T; g.N = C; g.K = T; g.lda = T; g.ldb = 3 * C; g.ldc = C;
  g.sA = (size_t)T * T; g.sB = (size_t)T * 3 * C; g.sC = (size_t)T * C; g.alpha = 1.0f;
  if (launch_bgemm(true, g, B, s)) return 1;
  }
  memset(&c, 0, sizeof c);
  c.kind = 3; c.x1 = O; c.c1 = C; c.wp = P + o.proj; c.bias = P + o.projb; c.cout = C; c.y = X; c.res = X;
  c.B = B; c.H = h; c.W = ww;
  return launch_conv(c, s);
}

// init_conv (model.py:90,135) and down1.conv1 (model.py:54) as one 5x5 convolution from the image (dm_kernels_init5.hip).
// LDS bytes it needs for an HxW image (0: geometry not covered -> the caller keeps the two-layer form).
size_t init5_lds(const dm_unet* m, int H, int W) {
  if (H < 16 || W < 16 || H % 8 || W % 16) return 0;      // (H >= 16: a tile position has at most one special row class)
  const int sets = 2 * (W == 16 ? 3 : 2);
  const size_t f = (size_t)((m->Cin * 240 + 3) & ~3) + (size_t)sets * (25 * m->Cin + 1) * 64 + 72;  // patch + weight sets + reduction words
  return f * sizeof(float) <= 150 * 1024 ? f * sizeof(float) : 0;
}

int launch_init5(const dm_unet* m, const float* P, const float* x, float* y, float* stats, int B, int H, int W, hipStream_t s) {
  Init5Args a;
  memset(&a, 0, sizeof a);
  a.x = x; a.wc = P + m->o_i5w; a.y = y; a.stats_out = stats;
  a.B = B; a.Cin = m->Cin; a.H = H; a.W = W; a.C = m->C;
  a.tiles_x = W / 16; a.tiles_per_img = (H / 8) * (W / 16);
  a.g_log = ilog2(m->C / DM_GROUPS); a.out_half = m->bf16;
  const size_t lds = init5_lds(m, H, W);
  if (!lds) return fail("init5: geometry %dx%d not covered", H, W);
  static bool attr_done[DM_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&init5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  // workgroups keep one tile position: grid.x = tiles_per_img x (images walked in parallel)
  const unsigned per_cu = lds > 80 * 1024 ? 1u : lds > 53 * 1024 ? 2u : 3u;  // (142 registers: three waves per SIMD)
  unsigned par = per_cu * cu_count(dev) / (unsigned)(a.tiles_per_img * (m->C / 64));
  if (par < 1) par = 1;
  if (par > (unsigned)B) par = (unsigned)B;
  const double pix = (double)B * H * W;
  // booked as the two layers it replaces: 9 Cin C + 9 C C multiply-adds per pixel; bytes: image in, activation out
  ProfScope ps(DM_K_INIT, s, 2.0 * pix * (9.0 * m->Cin * m->C + 9.0 * m->C * m->C), pix * (4.0 * m->Cin + (m->bf16 ? 2.0 : 4.0) * m->C));
  hipLaunchKernelGGL(init5_kernel, dim3(par * a.tiles_per_img, (unsigned)(m->C / 64)), dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Does block 0 run init5_kernel (init_conv composed with down1.conv1) on HxW images, and - guided sampling's doubled batch
// [x;x] of 2*dup_half images - once for both halves?
bool uses_init5(const dm_unet* m, int H, int W) {
  const int x6 = m->bf16 ? 0 : m->x6;
  if ((x6 & 8) || !m->init5) return false;
  const bool xs = H * W >= 16 && ((x6 & 4) && (H < 8 || W < 8));
  const int kd = (!m->bf16 && !xs && H >= 8 && W >= 8) ? 5 : 0;
  const ConvGeo g1 = conv_geo(kd, H, W, m->C);
  return !xs && g1.th_log == 3 && g1.tw_log == 4 && g1.ni == 1 && init5_lds(m, H, W) != 0;
}
bool shares_first_layers(const dm_unet* m, int B, int H, int W, int dup_half) {
  static const bool noshare = getenv("DM_DEBUG_NOSHARE") != nullptr;  // diagnostics only
  return dup_half > 0 && B == 2 * dup_half && !noshare && uses_init5(m, H, W);
}

// ------------------------------------------------------------------ any-geometry path (dm_kernels_generic.hip)
int launch_gconv(int kind, const float* x1, int c1, const float* x2, int c2, const float* wp, const float* bias, int cout,
                 float* y, int B, int H, int W, hipStream_t s) {
  if (kind < 0 || kind > 2) return fail("conv (any geometry): kind %d is not 3x3 / 4x4 stride 2 / transposed 4x4", kind);
  if (B <= 0 || H <= 0 || W <= 0) return fail("conv: empty input");
  if (c1 <= 0 || c1 % 8 || c2 % 8 || cout <= 0) return fail("conv (any geometry): input channel counts must be multiples of 8 (got %d+%d -> %d)", c1, c2, cout);
  if (kind == 1 && (H % 2 || W % 2)) return fail("conv: the stride-2 conv needs even H, W (got %dx%d)", H, W);
  GConvArgs a;
  memset(&a, 0, sizeof a);
  a.x1 = x1; a.x2 = c2 ? x2 : nullptr; a.wp = wp; a.bias = bias; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C1 = c1; a.C2 = c2; a.Cout = cout;
  a.OH = kind == 1 ? H / 2 : kind == 2 ? 2 * H : H;
  a.OW = kind == 1 ? W / 2 : kind == 2 ? 2 * W : W;
  a.Hb = kind == 1 ? H / 2 : H; a.Wb = kind == 1 ? W / 2 : W;
  a.tiles_x = (a.Wb + 7) / 8; a.tiles_y = (a.Hb + 7) / 8;
  a.nb32 = (cout + 31) / 32;
  const size_t nblk = (size_t)B * a.tiles_x * a.tiles_y;
  if (nblk > 0x7fffffffull) return fail("conv (any geometry): %zu tiles exceed the grid limit", nblk);
  const dim3 grid((unsigned)nblk, (unsigned)((cout + 63) / 64), kind == 2 ? 4u : 1u);
  const double taps = kind == 0 ? 9 : 16, cin = c1 + c2;
  ProfScope ps(DM_K_GENERIC_CONV, s, 2.0 * B * a.Hb * a.Wb * cout * taps * cin,
               4.0 * ((double)B * H * W * cin + (double)B * a.OH * a.OW * cout + taps * cin * cout));
  if (kind == 0) hipLaunchKernelGGL(gconv_kernel<0>, grid, dim3(256), 0, s, a);
  else if (kind == 1) hipLaunchKernelGGL(gconv_kernel<1>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gconv_kernel<2>, grid, dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

// x <- SiLU(GroupNorm8(x)) + temb in place: statistics kernel, then the elementwise kernel (model.py:54-62, 100-101)
int launch_gnorm(float* x, float* stats, const float* gamma, const float* beta, const float* temb, int temb_stride,
                 int temb_step_stride, const int* step_ptr, int step_off, int B, int HW, int C, hipStream_t s) {
  if (C % 8 || C <= 0) return fail("GroupNorm(8): channels must be a multiple of 8 (got %d)", C);
  if ((size_t)B * DM_GROUPS > 0x7fffffffull) return fail("GroupNorm: batch too large");
  ProfScope ps(DM_K_GENERIC_NORM, s, 12.0 * B * HW * C, 16.0 * B * HW * C);
  hipLaunchKernelGGL(gstats_kernel, dim3((unsigned)(B * DM_GROUPS)), dim3(256), 0, s, x, stats, HW, C);
  HIP_TRY(hipGetLastError());
  GNormArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.stats = stats; a.gamma = gamma; a.beta = beta; a.temb = temb; a.step_ptr = step_ptr; a.step_off = step_off;
  a.temb_stride = temb_stride; a.temb_step_stride = temb_step_stride; a.HW = HW; a.C = C;
  a.ppb = 16384 / C > 0 ? 16384 / C : 1;
  a.chunks = (HW + a.ppb - 1) / a.ppb;
  const size_t lds = (size_t)3 * C * sizeof(float);
  if (lds > 64 * 1024) return fail("GroupNorm: %d channels exceed the table size", C);
  hipLaunchKernelGGL(gnorm_kernel, dim3((unsigned)((size_t)B * a.chunks)), dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_gfinal(const float* u, float* stats, const float* gamma, const float* beta, const float* wt, const float* bias,
                  float* out, int B, int HW, int C, int Cout, hipStream_t s) {
  ProfScope ps(DM_K_FINAL, s, 2.0 * B * HW * C * Cout, 4.0 * B * HW * (2.0 * C + Cout));
  hipLaunchKernelGGL(gstats_kernel, dim3((unsigned)(B * DM_GROUPS)), dim3(256), 0, s, u, stats, HW, C);
  HIP_TRY(hipGetLastError());
  const int ppb = 64;
  const int chunks = (HW + ppb - 1) / ppb;
  const size_t lds = (size_t)(2 * C + Cout * C) * sizeof(float);
  if (lds > 160 * 1024) return fail("final_conv: %d x %d weights exceed the CU's LDS", Cout, C);
  if (lds > 48 * 1024) {
    static bool attr_done[DM_MAX_DEVICES] = {false};
    const int dev = current_device();
    if (!attr_done[dev]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gfinal_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL(gfinal_kernel, dim3((unsigned)((size_t)B * chunks)), dim3(256), lds, s, u, stats, gamma, beta, wt, bias, out, HW, C,
                     Cout, ppb, chunks);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_init_conv(const dm_unet* m, const float* P, const float* x, float* y, int B, int H, int W, hipStream_t s) {
  const int C = m->C;
  const size_t lds = (size_t)m->Cin * 9 * C * sizeof(float);
  if (lds > 160 * 1024) return fail("init_conv: %d x %d channels exceed the CU's LDS", m->Cin, C);
  if (lds > 48 * 1024) {
    static bool attr_done[DM_MAX_DEVICES] = {false};
    const int dev = current_device();
    if (!attr_done[dev]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&init_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done[dev] = true;
    }
  }
  const size_t total = (size_t)B * H * W * (C / 4);
  unsigned blocks = (unsigned)((total + 255) / 256 > 65536u * 4 ? 65536u * 4 : (total + 255) / 256);
  ProfScope ps(DM_K_INIT, s, 2.0 * B * H * W * C * 9 * m->Cin, 4.0 * B * H * W * (m->Cin + C));
  hipLaunchKernelGGL(init_conv_kernel, dim3(blocks), dim3(256), lds, s, x, P + m->o_initw, P + m->o_initb, y, B, m->Cin, H, W, C);
  HIP_TRY(hipGetLastError());
  return 0;
}

// model.py:135-151 for any H, W divisible by 8 and any C divisible by 8: the layer sequence of unet_body, each
// GroupNorm+SiLU(+temb) as its own pair of kernels, every convolution on gconv_kernel.
int unet_body_generic(const dm_unet* m, const float* P, const float* x, float* eps, const float* heads,
                      int heads_img_stride, int heads_step_stride, const int* step_ptr, int step_off,
                      const Workspace& w, int B, int H, int W, hipStream_t s) {
  const int C = m->C;
  if (m->bf16 || m->attn) return fail("internal: the any-geometry path is fp32 without attention");
  if (launch_init_conv(m, P, x, w.A0, B, H, W, s)) return 1;
  auto block = [&](int k, const float* in1, const float* in2, int h, int ww, bool up, float* out) -> int {
    const BlockOff& o = m->blk[k];
    if (launch_gconv(0, in1, C, in2, in2 ? C : 0, P + o.c1, P + o.c1b, C, w.T1, B, h, ww, s)) return 1;
    if (launch_gnorm(w.T1, w.SA, P + o.g1, P + o.b1, heads + (size_t)k * C, heads_img_stride, heads_step_stride, step_ptr, step_off,
                     B, h * ww, C, s)) return 1;
    if (launch_gconv(0, w.T1, C, nullptr, 0, P + o.c2, P + o.c2b, C, w.T2, B, h, ww, s)) return 1;
    if (launch_gnorm(w.T2, w.SA, P + o.g2, P + o.b2, nullptr, 0, 0, nullptr, 0, B, h * ww, C, s)) return 1;
    return launch_gconv(up ? 2 : 1, w.T2, C, nullptr, 0, P + o.tr, P + o.trb, C, out, B, h, ww, s);
  };
  if (block(0, w.A0, nullptr, H, W, false, w.D1)) return 1;
  if (block(1, w.D1, nullptr, H / 2, W / 2, false, w.D2)) return 1;
  if (block(2, w.D2, nullptr, H / 4, W / 4, false, w.D3)) return 1;
  const int hb = H / 8, wb = W / 8;
  if (launch_gconv(0, w.D3, C, nullptr, 0, P + m->o_b0, P + m->o_b0b, C, w.BT0, B, hb, wb, s)) return 1;
  if (launch_gnorm(w.BT0, w.SA, P + m->o_bg, P + m->o_bb, nullptr, 0, 0, nullptr, 0, B, hb * wb, C, s)) return 1;
  if (launch_gconv(0, w.BT0, C, nullptr, 0, P + m->o_b3, P + m->o_b3b, C, w.BT, B, hb, wb, s)) return 1;
  if (block(3, w.BT, w.D3, hb, wb, true, w.U1)) return 1;
  if (block(4, w.U1, w.D2, H / 4, W / 4, true, w.U2)) return 1;
  if (block(5, w.U2, w.D1, H / 2, W / 2, true, w.A0)) return 1;
  return launch_gfinal(w.A0, w.SA, P + m->o_fg, P + m->o_fb, P + m->o_fw, P + m->o_fwb, eps, B, H * W, C, m->Cout, s);
}

float* g_lowres_taps = nullptr;  // tests only (dm_debug_lowres_taps): where the fused kernel also writes its inner block outputs

// The fused low-resolution kernel (dm_kernels_lowres.hip) covers the 32x32 model with 64 channels in fp32 with the
// three-term split-bf16 weight forms; a property of the model and the image size only, never of the batch.
bool uses_lowres(const dm_unet* m, int H, int W) {
  return m->lowres && tuned_geometry(m, H, W) && !m->attn && m->C == 64 && H == 32 && W == 32 && (m->bf16 || (m->x6 & 31) == 7);
}

int launch_lowres(const dm_unet* m, const float* P, const float* t2, const float* st2, int pin2, float cnt2, const float* heads,
                  int heads_img_stride, int heads_step_stride, const int* step_ptr, int step_off, float* d2, float* u2, float* amax_u2,
                  int B, hipStream_t s) {
  LowresArgs a;
  memset(&a, 0, sizeof a);
  // wtot > 0: the pack is two fp16 planes of wtot elements with the float that undoes its power-of-two scale behind them
  auto layer = [&](size_t w, size_t bias, size_t gamma, size_t beta, bool norm, int temb_k, size_t wtot) {
    LrLayer L;
    L.w = reinterpret_cast<const bf16_t*>(P + w); L.bias = P + bias;
    L.gamma = norm ? P + gamma : nullptr; L.beta = norm ? P + beta : nullptr; L.temb_k = temb_k;
    L.wsc = wtot ? P + w + wtot : nullptr;
    return L;
  };
  const BlockOff &d2b = m->blk[1], &d3 = m->blk[2], &u1 = m->blk[3], &u2b = m->blk[4];
  // fp32 path: two fp16 terms per operand (*h packs; DM_X6 bit 7) or the three-term split-bf16 packs (*x); bf16 path: the
  // (hi, lo) pair packs - the same order, two terms
  const bool bh = m->bf16 != 0 && m->bf_h == 2;   // bf16 path, one half term per operand (lowres_fused_kernel<4>): the *h packs' first plane
  const bool bf = m->bf16 != 0 && !bh;            // bf16 path, bf16 operands with (hi, lo) weight pairs (<1>)
  const bool hf = bh || (!m->bf16 && (m->x6 & 128));
  const size_t C = (size_t)m->C, n3 = hf ? 9 * C * C : 0, n3c = 2 * n3, n4 = hf ? 16 * C * C : 0;
  a.t2 = t2; a.st2 = st2; a.pin2 = pin2; a.cnt2 = cnt2; a.g2 = P + d2b.g2; a.b2 = P + d2b.b2;
  a.d2t = layer(bf ? d2b.tr : hf ? d2b.trh : d2b.trx, d2b.trb, 0, 0, false, -1, n4);
  a.d3c1 = layer(bf ? d3.c1 : hf ? d3.c1h : d3.c1x, d3.c1b, d3.g1, d3.b1, true, 2, n3);
  a.d3c2 = layer(bf ? d3.c2 : hf ? d3.c2h : d3.c2x, d3.c2b, d3.g2, d3.b2, true, -1, n3);
  a.d3t = layer(bf ? d3.tr : hf ? d3.trh : d3.trx, d3.trb, 0, 0, false, -1, n4);
  a.b0 = layer(bf ? m->o_b0 : hf ? m->o_b0h : m->o_b0x, m->o_b0b, m->o_bg, m->o_bb, true, -1, n3);
  a.b3 = layer(bf ? m->o_b3 : hf ? m->o_b3h : m->o_b3x, m->o_b3b, 0, 0, false, -1, n3);
  a.u1c1 = layer(bf ? u1.c1 : hf ? u1.c1h : u1.c1x, u1.c1b, u1.g1, u1.b1, true, 3, n3c);
  a.u1c2 = layer(bf ? u1.c2 : hf ? u1.c2h : u1.c2x, u1.c2b, u1.g2, u1.b2, true, -1, n3);
  a.u1t = layer(bf ? u1.tr : hf ? u1.trh : u1.trx, u1.trb, 0, 0, false, -1, n4);
  a.u2c1 = layer(bf ? u2b.c1 : hf ? u2b.c1h : u2b.c1x, u2b.c1b, u2b.g1, u2b.b1, true, 4, n3c);
  a.u2c2 = layer(bf ? u2b.c2 : hf ? u2b.c2h : u2b.c2x, u2b.c2b, u2b.g2, u2b.b2, true, -1, n3);
  a.u2t = layer(bf ? u2b.tr : hf ? u2b.trh : u2b.trx, u2b.trb, 0, 0, false, -1, n4);
  a.heads = heads; a.heads_img_stride = heads_img_stride; a.heads_step_stride = heads_step_stride;
  a.step_ptr = step_ptr; a.step_off = step_off; a.d2 = d2; a.u2 = u2; a.B = B;
  a.taps = g_lowres_taps;
  a.amax_u2 = amax_u2;
#ifdef DM_STAMPS
  a.stamps = g_stamps;
#endif
  static bool attr_done[DM_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lowres_fused_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lowres_fused_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lowres_fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lowres_fused_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  unsigned grid = 2u * cu_count(dev);  // persistent: two workgroups per CU (60 KB of LDS each), images n, n + grid, ...
  if (grid > (unsigned)B) grid = (unsigned)B;
  // algorithmic work of the twelve layers: 25,231,360 multiply-adds per image
  ProfScope ps(DM_K_LOWRES, s, 2.0 * 25231360.0 * B, 4.0 * B * (2.0 * 16 * 16 * 64 + 2.0 * 8 * 8 * 64) + 6.0 * 12.6 * 64 * 64 * 9);
  if (bh) hipLaunchKernelGGL(lowres_fused_kernel<4>, dim3(grid), dim3(256), (size_t)lr::LDS_BYTES, s, a);
  else if (bf) hipLaunchKernelGGL(lowres_fused_kernel<1>, dim3(grid), dim3(256), (size_t)lr::LDS_BYTES, s, a);
  else if (hf) hipLaunchKernelGGL(lowres_fused_kernel<2>, dim3(grid), dim3(256), (size_t)lr::LDS_BYTES, s, a);
  else hipLaunchKernelGGL(lowres_fused_kernel<3>, dim3(grid), dim3(256), (size_t)lr::LDS_BYTES, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

// One denoiser evaluation (everything after the time embedding): model.py:135-151.
// heads: [.., 6, C] table; heads_img_stride = 6*C when every image has its own timestep
// (training-style [B] t) else 0; in the sampling loop the table is indexed by the step
// (step_ptr/step_off, stride 6*C per step, or 6*C*B with per-image heads).
// dup_half > 0: images [dup_half, 2 dup_half) of x are copies of [0, dup_half) (guided sampling's doubled batch): the first
// layers, which see no time / class embedding, are evaluated once and read by both halves.
int unet_body(const dm_unet* m, const float* P, const float* x, float* eps, const float* heads,
              int heads_img_stride, int heads_step_stride, const int* step_ptr, int step_off,
              const Workspace& w, int B, int H, int W, hipStream_t s, int dup_half = 0, const FinalStep* fstep = nullptr) {
  const int C = m->C;
  if (fstep && !fuses_step(m, H, W)) return fail("internal: fused sampler update requested for a path without it");
  if (!tuned_geometry(m, H, W))
    return unet_body_generic(m, P, x, eps, heads, heads_img_stride, heads_step_stride, step_ptr, step_off, w, B, H, W, s);
  // init_conv (model.py:135) is folded into down1.conv1's staging (no kernel, no HBM round trip)
  // (also in bf16: the stand-alone init kernel + a plain first layer take 343 + 163 us at B = 2048, the folded layer 420)
  const int x6 = m->bf16 ? 0 : m->x6;
  const bool fold_init = !(x6 & 8);  // (the split-bf16 3x3 kernel has no folded-init staging: stand-alone init kernel)
  if (!fold_init) {
    const size_t lds = (size_t)m->Cin * 9 * C * sizeof(float);
    const size_t total = (size_t)B * H * W * (C / 4);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 65536u * 4) blocks = 65536u * 4;
    ProfScope ps(DM_K_INIT, s, 2.0 * B * H * W * C * 9 * m->Cin, 4.0 * B * H * W * (m->Cin + C));
    if (m->bf16)
      hipLaunchKernelGGL(init_conv_bf16_kernel, dim3(blocks), dim3(256), lds, s, x, P + m->o_initw, P + m->o_initb,
                         reinterpret_cast<bf16_t*>(w.A0), B, m->Cin, H, W, C);
    else
      hipLaunchKernelGGL(init_conv_kernel, dim3(blocks), dim3(256), lds, s, x, P + m->o_initw, P + m->o_initb, w.A0, B, m->Cin, H, W, C);
    HIP_TRY(hipGetLastError());
  }
  ConvCall c;
  auto base = [&](int kind, const float* x1, int c1, const float* x2, int c2, size_t wp, size_t bias, float* y, float* so, int h, int ww) {
    memset(&c, 0, sizeof c);
    c.kind = kind; c.x1 = x1; c.c1 = c1; c.x2 = x2; c.c2 = c2; c.wp = P + wp; c.bias = P + bias; c.cout = C;
    c.y = y; c.stats_out = so; c.stats_cap = w.stats_floats; c.B = B; c.H = h; c.W = ww; c.bf16 = m->bf16;
  };
  auto with_tf = [&](const float* si, int kind_prod, int hp, int wp_, size_t g, size_t b) {
    const ConvGeo pg = conv_geo(kind_prod, hp, wp_, C);
    c.stats_in = si; c.pin = pg.pout; c.cnt_in = pg.cnt_out; c.gamma = P + g; c.beta = P + b;
  };
  // Block.forward (model.py:52-64); in: block input(s) at hxw
  // 3x3 convolutions: Winograd F(2x2,3x3) in fp32 from 8x8 up, the direct form below that and in bf16
  // (split-bf16 form, conv_x6_kernel: where m->x6 selects it)
  // (not below 16 pixels per image: 16-64 single-pixel images with their halos do not fit the CU's LDS as three planes)
  const int xmode = (x6 & 16) ? 2 : 1;  // bit 4: two-term splits (three MFMAs per product; about 7x the fp32 rounding error) - opt-in
  auto x3 = [&](int h, int ww) { return h * ww >= 16 && ((x6 & 8) || ((x6 & 4) && (h < 8 || ww < 8))); };
  auto k3 = [&](int h, int ww) { return (!m->bf16 && !x3(h, ww) && h >= 8 && ww >= 8) ? 5 : 0; };
  struct AmaxRef { const float* p; int n; };   // a tensor's maxima as its producer wrote them (null: none)
  const AmaxRef no_amax = {nullptr, 0};
  auto block = [&](int k, const float* in1, const float* in2, int h, int ww, bool up, float* out, float* out_stats,
                   bool convs_only, AmaxRef a1, AmaxRef a2, float* am_buf, AmaxRef* am_out) -> int {
    const BlockOff& o = m->blk[k];
    const int kd = k3(h, ww);
    const bool xs = x3(h, ww);
    // conv1 reads its inputs raw: two fp16 terms where the producers left their maxima (conv1_f16), else Winograd / direct
    const bool c1h = k != 0 && !xs && conv1_f16(m, h, ww) && a1.p && (!in2 || a2.p);
    const int k1 = c1h ? 0 : kd;
    base(k1, in1, C, in2, in2 ? C : 0, c1h ? o.c1h : xs ? o.c1x : kd == 5 ? o.c1w : o.c1, o.c1b, w.T1, w.SA, h, ww);
    c.x6 = c1h ? 3 : xs ? xmode : 0;
    if (c1h) { c.amax1 = a1.p; c.namax1 = a1.n; c.amax2 = in2 ? a2.p : nullptr; c.namax2 = in2 ? a2.n : 0; }
    if (k == 0 && fold_init) {
      c.init_x = x; c.init_w = P + m->o_initw; c.init_b = P + m->o_initb; c.init_cin = m->Cin;
    }
    // down1.conv1 after init_conv: ONE 5x5 convolution from the image where the 3x3 kernel's tiles are 8x16 pixels of one
    // image (so the consumer sees the same GroupNorm partial geometry)
    bool shared1 = false;  // conv1's output holds dup_half images only
    if (k == 0 && uses_init5(m, h, ww)) {
      shared1 = shares_first_layers(m, B, h, ww, dup_half);
      if (launch_init5(m, P, x, w.T1, w.SA, shared1 ? dup_half : B, h, ww, s)) return 1;
    } else if (launch_conv(c, s)) return 1;
    // conv2 reads SiLU(GroupNorm(conv1)) + temb: two fp16 terms, three bf16-rate MFMAs per product, direct form - 3/16 of the
    // fp32 MFMA time against Winograd's 4/9, and as exact; tiles of one image
    const bool c2h = conv2_f16(m, h, ww, shared1);
    const int k2 = c2h ? 0 : kd;
    base(k2, w.T1, C, nullptr, 0, c2h ? o.c2h : xs ? o.c2x : k2 == 5 ? o.c2w : o.c2, o.c2b, w.T2, w.SB, h, ww);
    c.x6 = c2h ? 3 : xs ? xmode : 0;  // (c2h implies h * w >= 64, where xs is off unless DM_X6 bit 3 is set)
    c.f16op = bf_half_layer(m, k, 2) ? m->bf_h : 0;
    c.in_wrap = shared1 ? dup_half : 0;
    with_tf(w.SA, k1, h, ww, o.g1, o.b1);
    c.temb = heads + (size_t)k * C; c.temb_stride = heads_img_stride; c.temb_step_stride = heads_step_stride;
    c.step_ptr = step_ptr; c.step_off = step_off;
    if (launch_conv(c, s)) return 1;
    if (convs_only) return 0;  // (the transform belongs to the fused low-resolution kernel)
    int kdn = (!m->bf16 && (h / 2) * (ww / 2) >= 256) ? 6 : 1;  // stride-2 conv: 128-pixel tiles from 16x16 outputs up
    const bool xt = up ? ((x6 & 2) && h * ww >= 16) : ((x6 & 1) && (h / 2) * (ww / 2) >= 16);
    // Small batches: when the 128-pixel tiles would leave half of the CUs without one, the split-bf16 stride-2 conv runs on
    // 64-pixel tiles (twice the workgroups, half the K-loop latency each).  Its output carries no statistics and its K order
    // (phase, tap, channel slab, split term) does not depend on the tile, so the result is bit-identical either way - the
    // one place where the launch shape may follow the batch size without touching per-image results.
    if (!up && xt && kdn == 6 && !out_stats && (size_t)B * ((h / 2) * (ww / 2) / 128) <= cu_count(current_device())) kdn = 1;
    // the stride-2 / transposed convolutions read SiLU(GroupNorm(.)): two fp16 terms and three MFMAs per product are as exact
    // there as the three-way bf16 split; tiles of one image only (the swapped epilogue)
    const ConvGeo gt = conv_geo(up ? 2 : kdn, h, ww, C);
    const bool xh = xt && xmode == 1 && (x6 & 32) && gt.ni == 1;
    base(up ? 2 : kdn, w.T2, C, nullptr, 0, xh ? o.trh : xt ? o.trx : o.tr, o.trb, out, out_stats, h, ww);
    c.x6 = xh ? 3 : xt ? xmode : 0;
    c.f16op = bf_half_layer(m, k, 1) ? m->bf_h : 0;
    with_tf(w.SB, k2, h, ww, o.g2, o.b2);
    if (am_out) *am_out = no_amax;
    if (am_buf && am_out && xt && gt.ni == 1 && !dbg_noswap()) {  // (the swapped epilogue writes them)
      c.amax_out = am_buf;
      *am_out = AmaxRef{am_buf, gt.pout * (C / 64) * 4};
    }
    return launch_conv(c, s);
  };
  AmaxRef aD1 = no_amax, aD2 = no_amax, aU1 = no_amax, aU2 = no_amax;
  if (block(0, w.A0, nullptr, H, W, false, w.D1, nullptr, false, no_amax, no_amax, w.AM[0], &aD1)) return 1;
  if (uses_lowres(m, H, W)) {
    // everything at and below 8x8 pixels - down2.transform ... up2.transform, twelve layers - in ONE launch with the
    // activations in LDS (dm_kernels_lowres.hip); chosen by geometry only, so results do not depend on the batch
    if (block(1, w.D1, nullptr, H / 2, W / 2, false, nullptr, nullptr, true, aD1, no_amax, nullptr, nullptr)) return 1;
    const ConvGeo pg = conv_geo(conv2_f16(m, H / 2, W / 2, false) ? 0 : k3(H / 2, W / 2), H / 2, W / 2, C);
    if (launch_lowres(m, P, w.T2, w.SB, pg.pout, pg.cnt_out, heads, heads_img_stride, heads_step_stride, step_ptr, step_off, w.D2, w.U2, w.AM[3], B, s))
      return 1;
    if (w.AM[3]) aU2 = AmaxRef{w.AM[3], 4};
    if (block(5, w.U2, w.D1, H / 2, W / 2, true, w.A0, w.SA, false, aU2, aD1, nullptr, nullptr)) return 1;
    const ConvGeo pf = conv_geo(2, H / 2, W / 2, C);
    return launch_final(w.A0, m->bf16, w.SA, pf.pout, pf.cnt_out, P + m->o_fg, P + m->o_fb, P + m->o_fw, P + m->o_fwb, eps, B, H, W, C,
                        m->Cout, s, fstep);
  }
  if (block(1, w.D1, nullptr, H / 2, W / 2, false, w.D2, m->attn ? w.SA : nullptr, false, aD1, no_amax, w.AM[1], &aD2)) return 1;
  if (m->attn) aD2 = no_amax;  // (the attention block rewrites D2)
  if (m->attn) {
    const ConvGeo pg = conv_geo((!m->bf16 && (H / 4) * (W / 4) >= 256) ? 6 : 1, H / 2, W / 2, C);  // down2.transform
    if (attention_block(m, P, 0, w.D2, w.SA, pg.pout, pg.cnt_out, w, B, H / 4, W / 4, H, W, s)) return 1;
  }
  if (block(2, w.D2, nullptr, H / 4, W / 4, false, w.D3, nullptr, false, aD2, no_amax, nullptr, nullptr)) return 1;
  // bottleneck (model.py:98-103,143)
  const int kb = k3(H / 8, W / 8);
  const bool xb = x3(H / 8, W / 8);
  base(kb, w.D3, C, nullptr, 0, xb ? m->o_b0x : kb == 5 ? m->o_b0w : m->o_b0, m->o_b0b, w.BT0, w.SA, H / 8, W / 8);
  c.x6 = xb ? xmode : 0;
  if (launch_conv(c, s)) return 1;
  base(kb, w.BT0, C, nullptr, 0, xb ? m->o_b3x : kb == 5 ? m->o_b3w : m->o_b3, m->o_b3b, w.BT, nullptr, H / 8, W / 8);
  c.x6 = xb ? xmode : 0;
  with_tf(w.SA, kb, H / 8, W / 8, m->o_bg, m->o_bb);
  if (launch_conv(c, s)) return 1;
  // up path: channel concat [upstream, skip] (model.py:146-148) handled by two input pointers
  if (block(3, w.BT, w.D3, H / 8, W / 8, true, w.U1, m->attn ? w.SA : nullptr, false, no_amax, no_amax, w.AM[2], &aU1)) return 1;
  if (m->attn) aU1 = no_amax;
  if (m->attn) {
    const ConvGeo pg = conv_geo(2, H / 8, W / 8, C);
    if (attention_block(m, P, 1, w.U1, w.SA, pg.pout, pg.cnt_out, w, B, H / 4, W / 4, H, W, s)) return 1;
  }
  if (block(4, w.U1, w.D2, H / 4, W / 4, true, w.U2, nullptr, false, aU1, aD2, w.AM[3], &aU2)) return 1;
  if (block(5, w.U2, w.D1, H / 2, W / 2, true, w.A0, w.SA, false, aU2, aD1, nullptr, nullptr)) return 1;
  {  // final_conv (model.py:111-115,151)
    const ConvGeo pg = conv_geo(2, H / 2, W / 2, C);
    if (launch_final(w.A0, m->bf16, w.SA, pg.pout, pg.cnt_out, P + m->o_fg, P + m->o_fb, P + m->o_fw, P + m->o_fwb, eps, B, H, W,
                     C, m->Cout, s, fstep)) return 1;
  }
  return 0;
}

// final_conv (model.py:111-115,151): GroupNorm + SiLU + 1x1 in one pass over the NHWC activation
int launch_final(const float* u, int bf16, const float* stats, int pin, float cnt, const float* gamma, const float* beta,
                 const float* wt, const float* bias, float* out, int B, int H, int W, int C, int Cout, hipStream_t s,
                 const FinalStep* fs) {
  if (H * W % 64) return fail("final_conv: H*W must be a multiple of 64 (got %dx%d)", H, W);
  FinalStep f;
  memset(&f, 0, sizeof f);
  if (fs) {
    if (bf16 || C != 64) return fail("internal: the fused sampler update needs the fp32 C = 64 final_conv");
    f = *fs;
  }
  const size_t lds = (size_t)(3 * C + Cout * C) * sizeof(float);
  ProfScope ps(DM_K_FINAL, s, 2.0 * B * H * W * C * Cout, 4.0 * B * H * W * (C + Cout));
  const int slabs = (H * W / 64) >= 8 ? 8 : (H * W / 64);  // measured at 32x32, B=1024: 1 -> 96 us, 8 -> 62 us, 16 -> 85 us
  const dim3 grid((unsigned)(B * (H * W / 64 / slabs)));
  if (bf16)
    hipLaunchKernelGGL(final_conv_bf16_kernel, grid, dim3(256), lds, s, reinterpret_cast<const bf16_t*>(u), stats, pin, cnt,
                       gamma, beta, wt, bias, out, B, H * W, C, Cout, slabs);
  else
    hipLaunchKernelGGL(final_conv_kernel, grid, dim3(256), lds, s, u, stats, pin, cnt, gamma, beta, wt, bias, out, B, H * W, C,
                       Cout, slabs, f);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_step(const float* x, const float* eps, const float* noise, int noise_idx, const float* coefs_dev,
                const int* step_ptr, int step_off, const float* cbyval, float* out, size_t count,
                int cfg, float guidance, hipStream_t s) {
  if (count % 4) return fail("sampler_step: element count must be a multiple of 4 (got %zu)", count);
  StepArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.eps = eps; a.noise = noise; a.coefs = coefs_dev; a.step_ptr = step_ptr; a.out = out;
  a.count = count; a.step_off = step_off; a.cfg = cfg; a.guidance = guidance; a.noise_idx = noise_idx;
  if (cbyval) for (int i = 0; i < 5; ++i) a.c[i] = cbyval[i];
  size_t blocks = (count / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  ProfScope ps(DM_K_STEP, s, 6.0 * count, 4.0 * count * (noise ? 4 : 3));
  hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// ------------------------------------------------------------------ static scratch (dm_unet_forward_noalloc)
// One file-scope device array per GPU the library runs on, held until the process ends: the
// workspace of the entry point that may not allocate.  Host code never uses the array's name as
// an address: the per-device address comes from hipGetSymbolAddress (noalloc_scratch()).
static __device__ __align__(256) unsigned char scratch[DM_NOALLOC_SCRATCH_BYTES];

namespace {
void* g_scratch_ptr[DM_MAX_DEVICES] = {nullptr};
std::mutex g_scratch_mu;  // the per-device lookup below may be reached from several host threads at once

// Address of THIS device's copy of `scratch` (looked up once per device, with that device current).
int noalloc_scratch(void** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= DM_MAX_DEVICES) return fail("device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail("hipGetSymbolAddress returned a null address for the static scratch");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace

// ================================================================== C ABI
extern "C" {

int dm_init(void) {
  void* p = nullptr;
  return noalloc_scratch(&p);
}

size_t dm_noalloc_scratch_bytes(void) { return (size_t)DM_NOALLOC_SCRATCH_BYTES; }

int dm_unet_forward_noalloc(dm_unet* m, const float* packed, const float* x, const int64_t* t_dev, int nt,
                            const float* cond_dev, float* eps, int B, int H, int W, void* stream) {
  if (!m || !packed || !x || !t_dev || !eps) return fail("dm_unet_forward_noalloc: null argument");
  if (check_geometry(m, B, H, W)) return 1;
  if (nt != 1 && nt != B) return fail("timesteps must be 0-dim/[1] or [B] (got %d for batch %d)", nt, B);
  const size_t need = carve(m, nullptr, B, H, W, 1).bytes;
  if (need > (size_t)DM_NOALLOC_SCRATCH_BYTES)
    return fail("dm_unet_forward_noalloc: batch %d of %dx%d needs %zu B of scratch, the static buffer holds %zu B; "
                "use dm_unet_forward with a caller-owned workspace", B, H, W, need, (size_t)DM_NOALLOC_SCRATCH_BYTES);
  void* base = nullptr;
  if (noalloc_scratch(&base)) return 1;
  // every workspace region is written by this call before it is read: nothing has to start at zero
  Workspace w = carve(m, base, B, H, W, 1);
  hipStream_t s = (hipStream_t)stream;
  if (launch_time_embed(m, packed, t_dev, nt, cond_dev, w.temb, w.heads, s)) return 1;
  return unet_body(m, packed, x, eps, w.heads, nt == 1 ? 0 : 6 * m->C, 0, nullptr, 0, w, B, H, W, s);
}

const char* dm_last_error(void) { return g_err.c_str(); }
int dm_version(void) { return DM_ABI_VERSION; }
#ifndef DM_BUILD_KEY
#define DM_BUILD_KEY "unkeyed"
#endif
const char* dm_build_key(void) { return DM_BUILD_KEY; }

int dm_unet_create(dm_unet** out, int in_channels, int out_channels, int model_channels, int time_emb_dim) {
  return dm_unet_create_ex(out, in_channels, out_channels, model_channels, time_emb_dim, 0);
}

int dm_unet_set_precision(dm_unet* m, int bf16) {
  if (!m) return fail("dm_unet_set_precision: null handle");
  if (bf16 && m->attn) return fail("the bf16 path does not cover the attention blocks");
  if (bf16 && !m->fast) return fail("the bf16 kernels need model_channels in {64,128,256,512} and in/out channels <= 4");
  m->bf16 = bf16 ? 1 : 0;
  for (auto& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
  m->graphs.clear();
  return 0;
}

int dm_unet_num_params(const dm_unet* m) { return m ? DM_UNET_NUM_PARAMS + (m->attn ? 12 : 0) : 0; }

int dm_unet_create_ex(dm_unet** out, int in_channels, int out_channels, int model_channels, int time_emb_dim,
                      int attention) {
  if (!out) return fail("dm_unet_create: null out");
  if (in_channels < 1 || in_channels > 64 || out_channels < 1 || out_channels > 64)
    return fail("in/out channels must be in 1..64 (got %d, %d)", in_channels, out_channels);
  // model.py:48-49,100,112: nn.GroupNorm(8, C) needs C divisible by 8 (torch raises a ValueError otherwise)
  if (model_channels < 8 || model_channels % 8 || model_channels > 2048)
    return fail("model_channels must be a multiple of 8 in 8..2048 (GroupNorm(8, C); got %d)", model_channels);
  const bool fast = (model_channels == 64 || model_channels == 128 || model_channels == 256 || model_channels == 512) &&
                    in_channels <= 4 && out_channels <= 4;
  if (attention && !fast) return fail("the (build-defined) attention blocks need model_channels in {64,128,256,512} and in/out channels <= 4");
  if (time_emb_dim < 4 || time_emb_dim % 2 || time_emb_dim > 1024)
    return fail("time_emb_dim must be even, in 4..1024 (got %d)", time_emb_dim);
  dm_unet* m = new dm_unet();
  m->Cin = in_channels; m->Cout = out_channels; m->C = model_channels; m->E = time_emb_dim;
  m->attn = attention ? 1 : 0;
  m->bf16 = 0;
  m->fast = fast ? 1 : 0;
  m->x6 = getenv("DM_X6") ? atoi(getenv("DM_X6")) & 511 : DM_X6_DEFAULT;
  m->init5 = getenv("DM_INIT5") ? (atoi(getenv("DM_INIT5")) != 0) : 1;
  m->lowres = getenv("DM_LOWRES") ? (atoi(getenv("DM_LOWRES")) != 0) : 1;
  m->bf_h = getenv("DM_BF16_F16") ? (atoi(getenv("DM_BF16_F16")) % 3) : 2;
  const int C = m->C, E = m->E;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += align_up(n, 64); return o; };
  m->o_freqs = take(E / 2);
  m->o_tm1w = take((size_t)E * E); m->o_tm1b = take(E); m->o_tm2w = take((size_t)E * E); m->o_tm2b = take(E);
  m->o_initw = take((size_t)C * m->Cin * 9); m->o_initb = take(C);
  for (int k = 0; k < 6; ++k) {
    const bool up = k >= 3;
    BlockOff& b = m->blk[k];
    b.tmw = take((size_t)C * E); b.tmb = take(C);
    b.c1 = take(conv_packed_floats(0, up ? 2 * C : C, C)); b.c1b = take(C);
    b.tr = take(conv_packed_floats(up ? 2 : 1, C, C)); b.trb = take(C);
    b.c2 = take(conv_packed_floats(0, C, C)); b.c2b = take(C);
    b.g1 = take(C); b.b1 = take(C); b.g2 = take(C); b.b2 = take(C);
    if (!fast) { b.c1w = b.c2w = b.c1x = b.c2x = b.trx = b.trh = b.c2h = b.c1h = 0; continue; }  // (Winograd / split-bf16 forms: tuned kernels only)
    b.c1w = take(conv_packed_floats(5, up ? 2 * C : C, C)); b.c2w = take(conv_packed_floats(5, C, C));
    b.c1x = take(conv_packed_floats(7, up ? 2 * C : C, C)); b.c2x = take(conv_packed_floats(7, C, C));
    b.trx = take(conv_packed_floats(up ? 9 : 8, C, C));
    b.trh = take(conv_packed_floats(up ? 14 : 13, C, C));
    b.c2h = take(conv_packed_floats(15, C, C));
    b.c1h = take(conv_packed_floats(15, up ? 2 * C : C, C));
  }
  if (fast) {
    m->o_b0w = take(conv_packed_floats(5, C, C)); m->o_b3w = take(conv_packed_floats(5, C, C));
    m->o_b0x = take(conv_packed_floats(7, C, C)); m->o_b3x = take(conv_packed_floats(7, C, C));
    m->o_b0h = take(conv_packed_floats(15, C, C)); m->o_b3h = take(conv_packed_floats(15, C, C));
    m->o_i5w = take((size_t)9 * (25 * m->Cin + 1) * C);
  } else {
    m->o_b0w = m->o_b3w = m->o_b0x = m->o_b3x = m->o_b0h = m->o_b3h = m->o_i5w = 0;
  }
  m->o_b0 = take(conv_packed_floats(0, C, C)); m->o_b0b = take(C);
  m->o_bg = take(C); m->o_bb = take(C);
  m->o_b3 = take(conv_packed_floats(0, C, C)); m->o_b3b = take(C);
  m->o_fg = take(C); m->o_fb = take(C);
  m->o_fw = take((size_t)m->Cout * C); m->o_fwb = take(m->Cout);
  if (m->attn)
    for (int k = 0; k < 2; ++k) {
      AttnOff& a = m->att[k];
      a.ng = take(C); a.nb = take(C);
      a.qkv = take(conv_packed_floats(3, C, 3 * C)); a.qkvb = take(3 * C);
      a.proj = take(conv_packed_floats(3, C, C)); a.projb = take(C);
    }
  m->total = off;
  *out = m;
  return 0;
}

void dm_unet_destroy(dm_unet* m) {
  if (!m) return;
  for (int d = 0; d < DM_MAX_DEVICES; ++d) {
    for (auto& st : m->split_streams[d]) if (st) (void)hipStreamDestroy(st);
    for (auto& ev : m->split_events[d]) if (ev) (void)hipEventDestroy(ev);
  }
  for (auto& g : m->graphs) {
    (void)hipGraphExecDestroy(g.exec);
    (void)hipGraphDestroy(g.graph);
  }
  delete m;
}

size_t dm_unet_packed_floats(const dm_unet* m) { return m ? m->total : 0; }

int dm_unet_pack_weights(dm_unet* m, const float* const* p, int nparams, const float* freqs_dev,
                         float* packed, void* stream) {
  if (!m || !p || !packed) return fail("dm_unet_pack_weights: null argument");
  if (nparams != dm_unet_num_params(m)) return fail("expected %d parameter tensors, got %d", dm_unet_num_params(m), nparams);
  for (int i = 0; i < nparams; ++i) if (!p[i]) return fail("parameter %d is null", i);
  hipStream_t s = (hipStream_t)stream;
  const int C = m->C, E = m->E;
  auto copy = [&](size_t dst, const float* src, size_t n) -> int {
    HIP_TRY(hipMemcpyAsync(packed + dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
  };
  if (freqs_dev) {
    if (copy(m->o_freqs, freqs_dev, E / 2)) return 1;
  } else {
    std::vector<float> f(E / 2);
    const float sc = (float)(log(10000.0) / (E / 2 - 1));
    for (int i = 0; i < E / 2; ++i) f[i] = expf((float)i * -sc);
    HIP_TRY(hipMemcpy(packed + m->o_freqs, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  int i = 0;
  if (copy(m->o_tm1w, p[i++], (size_t)E * E) || copy(m->o_tm1b, p[i++], E) ||
      copy(m->o_tm2w, p[i++], (size_t)E * E) || copy(m->o_tm2b, p[i++], E) ||
      copy(m->o_initw, p[i++], (size_t)C * m->Cin * 9) || copy(m->o_initb, p[i++], C)) return 1;
  const bool xf = m->fast && !m->bf16;  // Winograd / split-bf16 weight forms (fp32 tuned kernels)
  const bool hpk = xf || (m->fast && m->bf16 && m->bf_h == 2);  // two-fp16-term packs (bf16 path: the fused low-resolution kernel reads their first plane)
  if (m->fast) {  // init_conv o down1.conv1 as one 5x5 convolution (dm_kernels_init5.hip): p[4], p[5] = init_conv, p[8], p[9] = down1.conv1
    const size_t n1 = (size_t)9 * (25 * m->Cin + 1) * C;
    hipLaunchKernelGGL(pack_init5_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, p[4], p[5], p[8], p[9], m->Cin, C,
                       packed + m->o_i5w);
    HIP_TRY(hipGetLastError());
  }
  auto do_block = [&](int k) -> int {
    const bool up = k >= 3;
    const BlockOff& b = m->blk[k];
    if (copy(b.tmw, p[i++], (size_t)C * E) || copy(b.tmb, p[i++], C)) return 1;
    if (xf && (pack_conv(5, p[i], up ? 2 * C : C, C, packed + b.c1w, s) || pack_conv(7, p[i], up ? 2 * C : C, C, packed + b.c1x, s))) return 1;
    if (hpk && pack_conv(15, p[i], up ? 2 * C : C, C, packed + b.c1h, s)) return 1;
    if (pack_conv(0, p[i++], up ? 2 * C : C, C, packed + b.c1, s, m->bf16) || copy(b.c1b, p[i++], C)) return 1;
    if (xf && pack_conv(up ? 9 : 8, p[i], C, C, packed + b.trx, s)) return 1;
    if (hpk && pack_conv(up ? 14 : 13, p[i], C, C, packed + b.trh, s)) return 1;
    if (pack_conv(up ? 2 : 1, p[i++], C, C, packed + b.tr, s, bf_half_layer(m, k, 1) ? 2 : m->bf16) || copy(b.trb, p[i++], C)) return 1;
    if (xf && (pack_conv(5, p[i], C, C, packed + b.c2w, s) || pack_conv(7, p[i], C, C, packed + b.c2x, s))) return 1;
    if (hpk && pack_conv(15, p[i], C, C, packed + b.c2h, s)) return 1;
    if (pack_conv(0, p[i++], C, C, packed + b.c2, s, bf_half_layer(m, k, 2) ? 2 : m->bf16) || copy(b.c2b, p[i++], C)) return 1;
    if (copy(b.g1, p[i++], C) || copy(b.b1, p[i++], C) || copy(b.g2, p[i++], C) || copy(b.b2, p[i++], C)) return 1;
    return 0;
  };
  for (int k = 0; k < 3; ++k) if (do_block(k)) return 1;
  if (xf && (pack_conv(5, p[i], C, C, packed + m->o_b0w, s) || pack_conv(5, p[i + 4], C, C, packed + m->o_b3w, s) ||
                   pack_conv(7, p[i], C, C, packed + m->o_b0x, s) || pack_conv(7, p[i + 4], C, C, packed + m->o_b3x, s))) return 1;
  if (hpk && (pack_conv(15, p[i], C, C, packed + m->o_b0h, s) || pack_conv(15, p[i + 4], C, C, packed + m->o_b3h, s))) return 1;
  if (pack_conv(0, p[i++], C, C, packed + m->o_b0, s, m->bf16) || copy(m->o_b0b, p[i++], C) ||
      copy(m->o_bg, p[i++], C) || copy(m->o_bb, p[i++], C) ||
      pack_conv(0, p[i++], C, C, packed + m->o_b3, s, m->bf16) || copy(m->o_b3b, p[i++], C)) return 1;
  for (int k = 3; k < 6; ++k) if (do_block(k)) return 1;
  if (copy(m->o_fg, p[i++], C) || copy(m->o_fb, p[i++], C) ||
      copy(m->o_fw, p[i++], (size_t)m->Cout * C) || copy(m->o_fwb, p[i++], m->Cout)) return 1;
  if (m->attn)
    for (int k = 0; k < 2; ++k) {
      const AttnOff& a = m->att[k];
      if (copy(a.ng, p[i++], C) || copy(a.nb, p[i++], C) ||
          pack_conv(3, p[i++], C, 3 * C, packed + a.qkv, s) || copy(a.qkvb, p[i++], 3 * C) ||
          pack_conv(3, p[i++], C, C, packed + a.proj, s) || copy(a.projb, p[i++], C)) return 1;
    }
  if (i != dm_unet_num_params(m)) return fail("internal: consumed %d params", i);
  return 0;
}

size_t dm_unet_workspace_bytes(const dm_unet* m, int B, int H, int W, int nsteps) {
  if (!m || B <= 0 || H <= 0 || W <= 0) return 0;
  return carve(m, nullptr, B, H, W, nsteps < 1 ? 1 : nsteps).bytes;
}

int dm_time_embed(dm_unet* m, const float* packed, const int64_t* t_dev, int nt, const float* cond_dev,
                  float* temb_dev, float* heads_dev, void* stream) {
  if (!m || !packed || !t_dev || !heads_dev) return fail("dm_time_embed: null argument");
  if (nt <= 0) return fail("dm_time_embed: nt must be positive");
  return launch_time_embed(m, packed, t_dev, nt, cond_dev, temb_dev, heads_dev, (hipStream_t)stream);
}

int dm_unet_forward(dm_unet* m, const float* packed, const float* x, const int64_t* t_dev, int nt,
                    const float* cond_dev, float* eps, int B, int H, int W, void* ws, size_t ws_bytes,
                    void* stream) {
  if (!m || !packed || !x || !t_dev || !eps || !ws) return fail("dm_unet_forward: null argument");
  if (check_geometry(m, B, H, W)) return 1;
  if (nt != 1 && nt != B) return fail("timesteps must be 0-dim/[1] or [B] (got %d for batch %d)", nt, B);
  Workspace w = carve(m, ws, B, H, W, 1);
  if (ws_bytes < w.bytes) return fail("workspace too small: %zu < %zu bytes", ws_bytes, w.bytes);
  hipStream_t s = (hipStream_t)stream;
  if (launch_time_embed(m, packed, t_dev, nt, cond_dev, w.temb, w.heads, s)) return 1;
  return unet_body(m, packed, x, eps, w.heads, nt == 1 ? 0 : 6 * m->C, 0, nullptr, 0, w, B, H, W, s);
}

int dm_sampler_step(const float* x, const float* eps, const float* noise, const float* coef_host,
                    float* out, size_t count, void* stream) {
  if (!x || !eps || !coef_host || !out) return fail("dm_sampler_step: null argument");
  if (count == 0) return 0;
  return launch_step(x, eps, noise, 0, nullptr, nullptr, 0, coef_host, out, count, 0, 0.f, (hipStream_t)stream);
}

int dm_sampler_step_batched(const float* x, const float* eps, const float* noise, const float* coefs_dev,
                            float* out, int B, size_t count_per_image, void* stream) {
  if (!x || !eps || !coefs_dev || !out) return fail("dm_sampler_step_batched: null argument");
  if (B <= 0 || count_per_image == 0) return 0;
  if (count_per_image % 4) return fail("sampler_step: element count per image must be a multiple of 4 (got %zu)", count_per_image);
  StepArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.eps = eps; a.noise = noise; a.coefs = coefs_dev; a.out = out;
  a.count = (size_t)B * count_per_image; a.img_count = count_per_image;
  size_t blocks = (a.count / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

int dm_attention(const float* qkv_dev, float* o_dev, int B, int T, int C, void* stream) {
  if (!qkv_dev || !o_dev) return fail("dm_attention: null argument");
  if (B <= 0) return fail("dm_attention: empty batch");
  const int rc = launch_attention_fused(qkv_dev, o_dev, B, T, C, (hipStream_t)stream);
  if (rc < 0) return fail("dm_attention: the fused kernel takes T in {64,128,256} tokens and C in {64,128,256,512} (got T=%d, C=%d)", T, C);
  return rc;
}

int dm_final_conv(const float* u_dev, const float* stats_dev, int pin, float cnt_in, const float* gamma, const float* beta,
                  const float* w_dev, const float* bias_dev, float* out_dev, int B, int H, int W, int C, int Cout, void* stream) {
  if (!u_dev || !stats_dev || !gamma || !beta || !w_dev || !bias_dev || !out_dev) return fail("dm_final_conv: null argument");
  if (B <= 0 || pin <= 0) return fail("dm_final_conv: bad sizes");
  if (C % 64 || C > 512 || Cout < 1 || Cout > 4) return fail("dm_final_conv: C must be a multiple of 64 (<= 512), Cout in 1..4");
  return launch_final(u_dev, 0, stats_dev, pin, cnt_in, gamma, beta, w_dev, bias_dev, out_dev, B, H, W, C, Cout, (hipStream_t)stream);
}

int dm_sampler_coefs(const float* acp, const float* acp_prev, const float* pv, int T,
                     const int64_t* ts, int n, float eta, float* coefs) {
  if (!acp || !acp_prev || !pv || !ts || !coefs) return fail("dm_sampler_coefs: null argument");
  for (int i = 0; i < n; ++i) {
    const int64_t t = ts[i];
    if (t < 0 || t >= T) return fail("timestep %lld out of range [0,%d)", (long long)t, T);
    float* c = coefs + (size_t)i * 8;
    // fp32 arithmetic exactly as torch evaluates schedulers.py:118-128 on 0-dim tensors
    c[0] = sqrtf(1.0f - acp[t]);
    c[1] = sqrtf(acp[t]);
    c[2] = sqrtf(acp_prev[t]);
    c[3] = sqrtf(1.0f - acp_prev[t]);
    c[4] = sqrtf(pv[t]) * eta;
    c[5] = c[6] = c[7] = 0.f;
  }
  return 0;
}

// Into how many sub-batches does dm_sample cut the B images of a captured loop?  The sub-batches are independent sampling chains
// (per-image results do not depend on the batch an image rides in: kernel choice by geometry only), captured as PARALLEL
// branches of the graph on side streams: the latency-bound launches of one chain (a few workgroups per layer below 16x16, the
// fused low-resolution kernel with one workgroup per image) run beside the other chains' kernels, and the tail of every
// persistent launch is filled by the neighbour's head.  A scheduling decision only - results are bit-identical for any split
// (test_split_batches_are_bit_identical).  With guidance every sub-batch is its own doubled batch [x_k ; x_k].  Not in eager mode.
// Measured (tools/diag/r3_split_ab*.sh, DDIM-50, 32x32 C=64, same box, 1 -> 4 branches): B=128 -1.5 %, B=256 +11 %, B=512 +3.3 %,
// B=1024 +2.6 % (two branches +3.2 %), B=2048 +1.3 %, B=8192 +0.8 %; ancestral-1000 B=256 +9 %, B=64 (two branches of 32) -4 %;
// bf16 B=1024 +4.2 %; widened 64x64 C=256 B=256 +-0, with attention +2 %.  Hence: four branches once a branch still fills the
// chip's workgroup slots with its largest layer (>= 64 images of 32x32), none below.  Eight branches lose against four at every size
// (B=1024 -4 %, B=512 -10 %: the process has four hardware queues).  DM_SPLIT = k forces k <= 8 (read per call).
static int sample_split(const dm_unet* m, int B, int H, int W, int guided) {
  const char* env = getenv("DM_SPLIT");
  const int forced = env ? atoi(env) : 0;
  if (B < 2) return 1;
  int k = forced > 0 ? forced : ((size_t)B * H * W >= ((size_t)256 << 10) ? 4 : 1);
  if (k > DM_MAX_BRANCHES) k = DM_MAX_BRANCHES;
  while (k > 1 && B % k) --k;
  (void)m;
  return k < 1 ? 1 : k;
}

int dm_stream_create(void** stream_out) {
  if (!stream_out) return fail("dm_stream_create: null argument");
  hipStream_t st = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream_out = st;
  return 0;
}

int dm_stream_destroy(void* stream) {
  if (!stream) return 0;
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

int dm_sample_branches(const dm_unet* m, int B, int H, int W, int guided) {
  if (!m || B <= 0 || H <= 0 || W <= 0) return 0;
  return sample_split(m, B, H, W, guided);
}

size_t dm_sample_workspace_bytes(const dm_unet* m, int B, int H, int W, int n, int guided, int with_noise) {
  if (!m || B <= 0 || H <= 0 || W <= 0 || n <= 0) return 0;
  const int S0 = n < 50 ? n : 50;
  const size_t count = (size_t)B * m->Cin * H * W;
  size_t need = carve(m, nullptr, guided ? 2 * B : B, H, W, n, with_noise ? (size_t)S0 * count : 0).bytes;
  const int ns = sample_split(m, B, H, W, guided);
  if (ns > 1) {  // one carve per sub-batch, back to back
    const size_t sub = (size_t)ns * carve(m, nullptr, (guided ? 2 : 1) * (B / ns), H, W, n, with_noise ? (size_t)S0 * (count / ns) : 0).bytes;
    if (sub > need) need = sub;
  }
  return need;
}

int dm_sample(dm_unet* m, const float* packed, float* x, int B, int H, int W, const int64_t* ts,
              const float* coefs, int n, const float* noise, const float* cond, float guidance,
              void* ws, size_t ws_bytes, int use_graph, void* stream) {
  return dm_sample_chunked(m, packed, x, B, H, W, ts, coefs, n, noise, noise ? 1 : 0, nullptr, nullptr, cond, guidance, ws, ws_bytes,
                           use_graph, stream);
}

int dm_sample_chunked(dm_unet* m, const float* packed, float* x, int B, int H, int W, const int64_t* ts,
                      const float* coefs, int n, const float* noise_all, int with_noise, dm_chunk_fn chunk_fn, void* chunk_user,
                      const float* cond, float guidance, void* ws, size_t ws_bytes, int use_graph, void* stream) {
  if (!m || !packed || !x || !ts || !coefs || !ws) return fail("dm_sample: null argument");
  if (with_noise && !noise_all && !chunk_fn) return fail("dm_sample: with_noise needs the noise tensor or a chunk callback");
  if (!with_noise && noise_all) return fail("dm_sample: a noise tensor was given but with_noise is 0");
  const bool noise = with_noise != 0;
  if (m->Cin != m->Cout) return fail("sampling needs in_channels == out_channels (got %d, %d)", m->Cin, m->Cout);
  if (check_geometry(m, B, H, W)) return 1;
  if (n <= 0) return fail("dm_sample: number of steps must be positive");
  const int cfg = cond ? 1 : 0;
  const int Bd = cfg ? 2 * B : B;  // denoiser batch
  const int S0 = n < 50 ? n : 50;  // steps per graph (and per noise staging chunk)
  const size_t count = (size_t)B * m->Cin * H * W;
  hipStream_t s = (hipStream_t)stream;
  // sub-batches as parallel branches of the captured loop (sample_split); NS = 1: the whole batch on the caller's stream
  const int NS = (use_graph && s) ? sample_split(m, B, H, W, cfg) : 1;
  const int Bk = B / NS;                        // images per sub-batch (NS divides B)
  const int Bdk = Bd / NS;                      // its denoiser batch
  const size_t countk = count / NS;
  Workspace wsub[DM_MAX_BRANCHES];
  wsub[0] = carve(m, ws, Bdk, H, W, n, noise ? (size_t)S0 * countk : 0);
  for (int k = 1; k < NS; ++k) {
    wsub[k] = carve(m, reinterpret_cast<char*>(ws) + (size_t)k * wsub[0].bytes, Bdk, H, W, n, noise ? (size_t)S0 * countk : 0);
    // one set of per-step tables and one step counter for all branches (with guidance the heads are per image: each branch its own)
    if (!cfg) wsub[k].heads = wsub[0].heads;
    wsub[k].temb = wsub[0].temb; wsub[k].coefs = wsub[0].coefs; wsub[k].ts = wsub[0].ts; wsub[k].step = wsub[0].step;
  }
  const Workspace& w = wsub[0];
  if (ws_bytes < (size_t)NS * w.bytes) return fail("workspace too small: %zu < %zu bytes (dm_sample_workspace_bytes)", ws_bytes, (size_t)NS * w.bytes);
  hipStream_t sk[DM_MAX_BRANCHES] = {s};
  hipEvent_t* ev = nullptr;
  if (NS > 1) {
    const int dev = current_device();
    for (int k = 1; k < NS; ++k) {
      if (!m->split_streams[dev][k - 1]) HIP_TRY(hipStreamCreateWithFlags(&m->split_streams[dev][k - 1], hipStreamNonBlocking));
      sk[k] = m->split_streams[dev][k - 1];
    }
    for (int k = 0; k < DM_MAX_BRANCHES; ++k)
      if (!m->split_events[dev][k]) HIP_TRY(hipEventCreateWithFlags(&m->split_events[dev][k], hipEventDisableTiming));
    ev = m->split_events[dev];
  }
  // Every operand the step kernels read is staged into the workspace, so a captured graph holds
  // workspace addresses only and is reusable whatever buffers the next call brings.
  for (int k = 0; k < NS; ++k) HIP_TRY(hipMemcpyAsync(wsub[k].xcur, x + (size_t)k * countk, countk * sizeof(float), hipMemcpyDefault, s));
  HIP_TRY(hipMemcpyAsync(w.ts, ts, (size_t)n * sizeof(int64_t), hipMemcpyDefault, s));
  HIP_TRY(hipMemcpyAsync(w.coefs, coefs, (size_t)n * 8 * sizeof(float), hipMemcpyDefault, s));
  if (cfg)  // cond = [B conditional rows ; B null rows]: branch k takes its Bk rows of each half
    for (int k = 0; k < NS; ++k)
      for (int hf = 0; hf < 2; ++hf)
        HIP_TRY(hipMemcpyAsync(wsub[k].cond + (size_t)hf * Bk * m->E, cond + ((size_t)hf * B + (size_t)k * Bk) * m->E, (size_t)Bk * m->E * sizeof(float),
                               hipMemcpyDefault, s));
  // Time embeddings of all n steps up front: they depend on t only.  With
  // guidance every image has its own embedding (time + class): computed per step for 2B rows.
  if (!cfg && launch_time_embed(m, packed, w.ts, n, nullptr, nullptr, w.heads, s)) return 1;
  // (guidance: the time MLP of all n steps once - it depends on t only -; per step only the heads of time + class embedding)
  if (cfg && launch_time_embed(m, packed, w.ts, n, nullptr, w.temb, w.heads, s)) return 1;
  const int hstep = 6 * m->C;

  // one step of sub-batch k on its stream; step index = *step_ptr + step_off (tables); `local` = index into the staged noise chunk
  auto one_step = [&](int k, const int* step_ptr, int step_off, int local) -> int {
    const Workspace& wk = wsub[k];
    hipStream_t st = sk[k];
    float* xc = wk.xcur;
    if (!cfg) {
      if (fuses_step(m, H, W)) {  // the update rides in final_conv's store: x is read and rewritten in place, element by element
        FinalStep fsx;
        memset(&fsx, 0, sizeof fsx);
        fsx.x = xc; fsx.noise = noise ? wk.noise : nullptr; fsx.coefs = wk.coefs; fsx.step_ptr = step_ptr; fsx.xout = xc;
        fsx.count = countk; fsx.step_off = step_off; fsx.noise_idx = local;
        return unet_body(m, packed, xc, wk.eps, wk.heads, 0, hstep, step_ptr, step_off, wk, Bdk, H, W, st, 0, &fsx);
      }
      if (unet_body(m, packed, xc, wk.eps, wk.heads, 0, hstep, step_ptr, step_off, wk, Bdk, H, W, st)) return 1;
      return launch_step(xc, wk.eps, noise ? wk.noise : nullptr, local, wk.coefs, step_ptr, step_off, nullptr, xc, countk, 0, 0.f, st);
    }
    // doubled batch [x;x] with embeddings [cond ; null] (build-defined CFG)
    // (when block 0 is evaluated once for both halves, only x itself is read: no duplicate needed)
    const bool shared = shares_first_layers(m, Bdk, H, W, Bk);
    if (!shared) {
      size_t blocks = (countk + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(dup_kernel, dim3((unsigned)blocks), dim3(256), 0, st, xc, wk.xx, countk);
      HIP_TRY(hipGetLastError());
    }
    static const bool no_table = getenv("DM_DEBUG_NO_TEMB_TABLE") != nullptr;  // diagnostics only: the time MLP per entry and step, as before
    if (launch_time_embed(m, packed, wk.ts, Bdk, wk.cond, nullptr, wk.heads, st, step_ptr, step_off, 0, no_table ? nullptr : wk.temb)) return 1;
    if (unet_body(m, packed, shared ? xc : wk.xx, wk.eps, wk.heads, hstep, 0, nullptr, 0, wk, Bdk, H, W, st, Bk)) return 1;
    return launch_step(xc, wk.eps, noise ? wk.noise : nullptr, local, wk.coefs, step_ptr, step_off, nullptr, xc, countk, 1, guidance, st);
  };
  // noise of steps [first, first + cnt) -> staging area (noise: [n][count], device or host memory)
  // The chunk callback runs on the calling thread BEFORE the chunk is enqueued (everything before `first` already is): it
  // is the per-chunk progress tick (sampler.py:65-66 ticks per step) and, when the caller draws the noise chunk by chunk,
  // the provider of steps [first, first + cnt) - the draw of chunk k+1 then overlaps the GPU's work on chunk k.
  auto stage_noise = [&](int first, int cnt) -> int {
    const float* src = noise_all ? noise_all + (size_t)first * count : nullptr;
    if (chunk_fn) {
      const float* got = chunk_fn(chunk_user, first, cnt);
      if (noise && !noise_all) {
        if (!got) return fail("dm_sample: the chunk callback returned no noise for steps [%d, %d)", first, first + cnt);
        src = got;
      }
    }
    if (!noise) return 0;
    if (NS == 1) {
      HIP_TRY(hipMemcpyAsync(w.noise, src, (size_t)cnt * count * sizeof(float), hipMemcpyDefault, s));
      return 0;
    }
    // sub-batch k takes columns [k countk, (k + 1) countk) of every step's row: [cnt][count] -> [cnt][countk]
    for (int k = 0; k < NS; ++k)
      HIP_TRY(hipMemcpy2DAsync(wsub[k].noise, countk * sizeof(float), src + (size_t)k * countk, count * sizeof(float), countk * sizeof(float),
                               (size_t)cnt, hipMemcpyDefault, s));
    return 0;
  };

  if (!use_graph) {
    for (int i = 0; i < n; ++i) {
      if (i % S0 == 0 && stage_noise(i, n - i < S0 ? n - i : S0)) return 1;
      if (one_step(0, nullptr, i, i % S0)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(x, w.xcur, count * sizeof(float), hipMemcpyDefault, s));
    return 0;
  }
  if (!s) return fail("dm_sample: graph capture needs a non-default stream");
  hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(64), 0, s, w.step, 0);
  HIP_TRY(hipGetLastError());
  auto get_graph = [&](int S, hipGraphExec_t* exec) -> int {
    for (auto& g : m->graphs)
      if (g.packed == packed && g.ws == ws && g.B == B && g.H == H && g.W == W && g.S == S && g.n == n &&
          g.cfg == cfg && g.noise == (noise ? 1 : 0) && g.guidance == guidance && g.bf16 == m->bf16 && g.nsplit == NS) {
        *exec = g.exec;
        return 0;
      }
    GraphEntry e;
    e.packed = packed; e.ws = ws; e.B = B; e.H = H; e.W = W; e.S = S; e.n = n; e.cfg = cfg; e.noise = noise ? 1 : 0;
    e.guidance = guidance; e.bf16 = m->bf16; e.nsplit = NS;
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    // fork: the side streams join the capture behind the caller's stream; every branch then runs its S steps on its own
    if (NS > 1) {
      if (hipEventRecord(ev[0], s) != hipSuccess) rc = fail("dm_sample: fork event");
      for (int k = 1; k < NS && !rc; ++k)
        if (hipStreamWaitEvent(sk[k], ev[0], 0) != hipSuccess) rc = fail("dm_sample: fork wait");
    }
    for (int k = 0; k < NS && !rc; ++k)
      for (int i = 0; i < S && !rc; ++i) rc = one_step(k, w.step, i, i);
    // join: the step counter advances when every branch is through
    for (int k = 1; k < NS && !rc; ++k) {
      if (hipEventRecord(ev[k], sk[k]) != hipSuccess || hipStreamWaitEvent(s, ev[k], 0) != hipSuccess) rc = fail("dm_sample: join");
    }
    if (!rc) {
      hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, w.step, S);
      if (hipGetLastError() != hipSuccess) rc = fail("advance_step launch failed");
    }
    hipError_t ce = hipStreamEndCapture(s, &e.graph);
    if (rc) { if (ce == hipSuccess && e.graph) (void)hipGraphDestroy(e.graph); return 1; }
    HIP_TRY(ce);
    HIP_TRY(hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0));
    if (m->graphs.size() >= 8) {
      (void)hipGraphExecDestroy(m->graphs.front().exec);
      (void)hipGraphDestroy(m->graphs.front().graph);
      m->graphs.erase(m->graphs.begin());
    }
    m->graphs.push_back(e);
    *exec = e.exec;
    return 0;
  };
  hipGraphExec_t ex;
  if (n / S0 > 0) {
    if (get_graph(S0, &ex)) return 1;
    for (int r = 0; r < n / S0; ++r) {
      if (stage_noise(r * S0, S0)) return 1;
      HIP_TRY(hipGraphLaunch(ex, s));
    }
  }
  if (n % S0) {
    if (get_graph(n % S0, &ex)) return 1;
    if (stage_noise(n - n % S0, n % S0)) return 1;
    HIP_TRY(hipGraphLaunch(ex, s));
  }
  for (int k = 0; k < NS; ++k) HIP_TRY(hipMemcpyAsync(x + (size_t)k * countk, wsub[k].xcur, countk * sizeof(float), hipMemcpyDefault, s));
  return 0;
}

int dm_unet_profile(dm_unet* m, const float* packed, float* x, const int64_t* t_dev, const float* coef_host,
                    int B, int H, int W, void* ws, size_t ws_bytes, int reps, void* stream,
                    int nk, int* counts, float* ms, double* flops, double* bytes) {
  if (!m || !packed || !x || !t_dev || !coef_host || !ws || !counts || !ms || !flops || !bytes)
    return fail("dm_unet_profile: null argument");
  if (nk != DM_K_COUNT) return fail("dm_unet_profile: the result arrays hold %d entries, this library reports DM_K_COUNT = %d kernel classes", nk, (int)DM_K_COUNT);
  if (check_geometry(m, B, H, W)) return 1;
  if (m->Cin != m->Cout) return fail("dm_unet_profile needs in_channels == out_channels");
  Workspace w = carve(m, ws, B, H, W, 1);
  if (ws_bytes < w.bytes) return fail("workspace too small: %zu < %zu bytes", ws_bytes, w.bytes);
  hipStream_t s = (hipStream_t)stream;
  if (fuses_step(m, H, W)) HIP_TRY(hipMemcpyAsync(w.coefs, coef_host, 8 * sizeof(float), hipMemcpyDefault, s));
  Prof prof;
  g_prof = &prof;
  int rc = 0;
  for (int r = 0; r < reps && !rc; ++r) {
    rc = launch_time_embed(m, packed, t_dev, 1, nullptr, w.temb, w.heads, s);
    // the update goes out-of-place (into the workspace) so repeated profiling passes see the same input
    if (!rc && fuses_step(m, H, W)) {  // as dm_sample runs it: the update inside final_conv
      FinalStep fsx;
      memset(&fsx, 0, sizeof fsx);
      fsx.x = x; fsx.coefs = w.coefs; fsx.xout = w.xx; fsx.count = (size_t)B * m->Cin * H * W;
      rc = unet_body(m, packed, x, w.eps, w.heads, 0, 0, nullptr, 0, w, B, H, W, s, 0, &fsx);
      continue;
    }
    if (!rc) rc = unet_body(m, packed, x, w.eps, w.heads, 0, 0, nullptr, 0, w, B, H, W, s);
    if (!rc) rc = launch_step(x, w.eps, nullptr, 0, nullptr, nullptr, 0, coef_host, w.xx, (size_t)B * m->Cin * H * W, 0, 0.f, s);
  }
  g_prof = nullptr;
  hipError_t se = hipStreamSynchronize(s);
  for (int k = 0; k < DM_K_COUNT; ++k) { counts[k] = 0; ms[k] = 0.f; flops[k] = prof.flops[k]; bytes[k] = prof.bytes[k]; }
  for (auto& r : prof.recs) {
    float t = 0.f;
    if (se == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { counts[r.cls]++; ms[r.cls] += t; }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  if (rc) return 1;
  HIP_TRY(se);
  return 0;
}

int dm_postprocess(const float* x_dev, float* out_f32_dev, unsigned char* out_u8_dev, size_t count, void* stream) {
  if (!x_dev || (!out_f32_dev && !out_u8_dev)) return fail("dm_postprocess: null argument");
  if (count == 0) return 0;
  size_t blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_dev, out_f32_dev, out_u8_dev, count);
  HIP_TRY(hipGetLastError());
  return 0;
}

int dm_class_embed(const float* table_dev, int num_rows, int E, const float* w1, const float* b1, const float* w2,
                   const float* b2, const int64_t* labels_dev, int n, float* out_dev, void* stream) {
  if (!table_dev || !w1 || !b1 || !w2 || !b2 || !labels_dev || !out_dev) return fail("dm_class_embed: null argument");
  if (n <= 0 || E <= 0 || E > 4096 || num_rows <= 0) return fail("dm_class_embed: bad sizes");
  hipLaunchKernelGGL(class_embed_kernel, dim3((unsigned)n), dim3(256), 2 * E * sizeof(float), (hipStream_t)stream,
                     table_dev, w1, b1, w2, b2, labels_dev, num_rows, E, out_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

#ifdef DM_STAMPS
void dm_debug_set_stamps(unsigned long long* p) { g_stamps = p; }
#endif

void dm_debug_lowres_taps(float* taps_dev) { g_lowres_taps = taps_dev; }

// tests: the maxima buffers of the NEXT dm_conv2d call (kinds 13-15; see ConvArgs.amax1 / amax_out), cleared by that call
// (per thread: a dm_conv2d call of ANOTHER thread - the test suite's co-runner - must not consume them)
static thread_local const float* g_dbg_amax1 = nullptr; static thread_local const float* g_dbg_amax2 = nullptr;
static thread_local int g_dbg_n1 = 0, g_dbg_n2 = 0; static thread_local float* g_dbg_amax_out = nullptr;
void dm_debug_conv_amax(const float* amax1, int namax1, const float* amax2, int namax2, float* amax_out) {
  g_dbg_amax1 = amax1; g_dbg_n1 = namax1; g_dbg_amax2 = amax2; g_dbg_n2 = namax2; g_dbg_amax_out = amax_out;
}

size_t dm_conv_packed_floats(int kind, int cin, int cout) { return conv_packed_floats(kind, cin, cout); }

int dm_pack_conv_weight(int kind, const float* w_dev, int cin, int cout, float* packed_dev, void* stream) {
  if (!w_dev || !packed_dev) return fail("dm_pack_conv_weight: null argument");
  if (kind < 0 || kind == 4 || kind > 15) return fail("unknown conv kind %d", kind);
  return pack_conv(kind, w_dev, cin, cout, packed_dev, (hipStream_t)stream);
}

// kinds 7, 8, 9: kinds 0, 1 (on 128-pixel tiles from 256 output pixels up, as the model does) and 2 in the split-bf16 form
// kinds 13, 14: kinds 8, 9 with the operands as two fp16 terms (inputs behind a GroupNorm only)
static int x6_base_kind(int kind, int H, int W) { return (kind == 7 || kind == 15) ? 0 : (kind == 9 || kind == 14) ? 2 : ((H / 2) * (W / 2) >= 256 ? 6 : 1); }

int dm_conv_stats_partials(int kind, int H, int W) {
  if (kind < 0 || kind == 4 || (kind > 9 && kind < 13) || kind > 15 || !is_pow2(H) || !is_pow2(W)) return 0;
  if (kind >= 7) kind = x6_base_kind(kind, H, W);
  return conv_geo(kind, H, W, 64).pout;
}

int dm_groupnorm_silu(float* x_dev, const float* gamma, const float* beta, const float* temb, int temb_stride,
                      float* stats_dev, int B, int H, int W, int C, void* stream) {
  if (!x_dev || !gamma || !beta || !stats_dev) return fail("dm_groupnorm_silu: null argument");
  if (B <= 0 || H <= 0 || W <= 0) return fail("dm_groupnorm_silu: empty input");
  return launch_gnorm(x_dev, stats_dev, gamma, beta, temb, temb_stride, 0, nullptr, 0, B, H * W, C, (hipStream_t)stream);
}

int dm_conv2d(int kind, const float* x1, int c1, const float* x2, int c2, const float* wp, const float* bias,
              int cout, const float* stats_in, int pin, float cnt_in, const float* gamma, const float* beta,
              const float* temb, int temb_stride, float* y, float* stats_out, int B, int H, int W, void* stream) {
  if (!x1 || !wp || !bias || !y) return fail("dm_conv2d: null argument");
  if (c2 && !x2) return fail("dm_conv2d: c2 > 0 but x2 is null");
  if (stats_in && (!gamma || !beta || pin <= 0)) return fail("dm_conv2d: input transform needs gamma, beta, pin");
  ConvCall c;
  memset(&c, 0, sizeof c);
  c.kind = kind; c.x1 = x1; c.c1 = c1; c.x2 = c2 ? x2 : nullptr; c.c2 = c2; c.wp = wp; c.bias = bias; c.cout = cout;
  c.stats_in = stats_in; c.pin = pin; c.cnt_in = cnt_in; c.gamma = gamma; c.beta = beta;
  c.temb = temb; c.temb_stride = temb_stride; c.y = y; c.stats_out = stats_out; c.B = B; c.H = H; c.W = W;
  if (kind >= 10 && kind <= 12) {  // any-geometry kernels: plain convolution (the GroupNorm transform is dm_groupnorm_silu)
    if (stats_in || stats_out || temb) return fail("dm_conv2d: kinds 10-12 take no fused GroupNorm transform / statistics");
    return launch_gconv(kind - 10, x1, c1, c2 ? x2 : nullptr, c2, wp, bias, cout, y, B, H, W, (hipStream_t)stream);
  }
  if (kind >= 7 && kind <= 9) { c.x6 = 1; c.kind = x6_base_kind(kind, H, W); }
  if (kind >= 13 && kind <= 15) { c.x6 = 3; c.kind = x6_base_kind(kind, H, W); }
  if (kind >= 7 && kind != 10 && kind != 11 && kind != 12) {
    c.amax1 = g_dbg_amax1; c.namax1 = g_dbg_n1; c.amax2 = g_dbg_amax2; c.namax2 = g_dbg_n2; c.amax_out = g_dbg_amax_out;
  }
  g_dbg_amax1 = g_dbg_amax2 = nullptr; g_dbg_n1 = g_dbg_n2 = 0; g_dbg_amax_out = nullptr;
  return launch_conv(c, (hipStream_t)stream);
}

}  // extern "C"
