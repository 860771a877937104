/* ORACLE -- test infrastructure only, never shipped, never on the product path.
 *
 * CPU reference backend of the C-ABI (include/semivl_hip.h) for the pixel-loss / label / metric / optimizer families:
 * the SAME entry points and argument conventions as libsemivl_hip.so implemented in plain C on host pointers
 * (`stream` is ignored), so that the boundary itself -- argument order, layouts, error convention, integer exactness --
 * can be exercised in a container without a GPU and so that the GPU tests can hold the HIP kernels
 * against a second, independent implementation of the same entry point on the same inputs.
 * Each function restates the reference computation it stands for:
 *   svl_softmax_max_f32      pred.softmax(dim=1).max(dim=1)                         semivl.py:232,252
 *   svl_cutmix_*             mask[box == 1] = mask_mix[box == 1]                    utils/train_utils.py:19-27
 *   svl_ce_fused_f32 (+finalize, semivl_gscale / semivl_loss, conf_avg_factor, conf_ratio_f32)
 *                            CE(ignore 255) / CE(none) * confidence weight / mc CE  semivl.py:52-58,267-323,
 *                                                                                   utils/train_utils.py:30-49
 *   svl_ce_up_fused_f32, svl_softmax_max_up_f32 (+ svl_ce_up_num_blocks)
 *                            the same on logits at the head's resolution: F.interpolate(bilinear) -> loss -> its backward
 *                                                                                   vlg_head.py:247, builder.py:93-97
 *   svl_maskclip_labels      upsample -> softmax(100 x) -> max -> threshold         model/vlm.py:100-109
 *   svl_concept_max_f32      per-class max over concept channels                    model/text_embeddings.py:188-193
 *   svl_iou_hist_i64         intersectionAndUnion                                   third_party/unimatch/util/utils.py:91-103
 *   svl_adamw_step           torch.optim.AdamW on the flat arena                    semivl.py:123-125,328
 * Built by __graft_entry__.build() with gcc into oracle/_ref/libsemivl_cpu.so (git-ignored).  Only tests load it. */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/semivl_hip.h"

static __thread char g_err[512];
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return SVL_ERR_INVALID_ARG;
}
#define CHECK(c, ...) do { if (!(c)) return fail(__VA_ARGS__); } while (0)

int svl_version(void) { return 300; }
int svl_last_error(char* buf, size_t len) {
  const size_t n = strlen(g_err);
  if (buf && len > 0) {
    const size_t c = n < len - 1 ? n : len - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)n;
}

int svl_fill_f32(float* p, float v, int64_t n, svl_stream_t s) {
  (void)s;
  CHECK(p && n > 0, "svl_fill_f32: bad args");
  for (int64_t i = 0; i < n; ++i) p[i] = v;
  return SVL_OK;
}

int svl_softmax_max_f32(const float* logits, int B, int N, int64_t HW, float* conf, int64_t* label, svl_stream_t s) {
  (void)s;
  CHECK(logits && conf && label && B > 0 && N > 0 && HW > 0, "svl_softmax_max_f32: bad args");
  for (int b = 0; b < B; ++b)
    for (int64_t p = 0; p < HW; ++p) {
      const float* x = logits + (int64_t)b * N * HW + p;
      float m = -INFINITY;
      int idx = 0;
      for (int c = 0; c < N; ++c)
        if (x[(int64_t)c * HW] > m) { m = x[(int64_t)c * HW]; idx = c; }   /* first max wins (torch.max) */
      float sum = 0.f;
      for (int c = 0; c < N; ++c) sum += expf(x[(int64_t)c * HW] - m);
      conf[(int64_t)b * HW + p] = 1.f / sum;
      label[(int64_t)b * HW + p] = idx;
    }
  return SVL_OK;
}

int svl_cutmix_f32(float* out, const float* a, const float* b, const float* box, int B, int C, int64_t HW, svl_stream_t s) {
  (void)s;
  CHECK(out && a && b && box && B > 0 && C > 0 && HW > 0, "svl_cutmix_f32: bad args");
  for (int bi = 0; bi < B; ++bi)
    for (int c = 0; c < C; ++c)
      for (int64_t p = 0; p < HW; ++p) {
        const int64_t i = ((int64_t)bi * C + c) * HW + p;
        out[i] = box[(int64_t)bi * HW + p] == 1.f ? b[i] : a[i];
      }
  return SVL_OK;
}
int svl_cutmix_i64(int64_t* out, const int64_t* a, const int64_t* b, const float* box, int B, int64_t HW, svl_stream_t s) {
  (void)s;
  CHECK(out && a && b && box && B > 0 && HW > 0, "svl_cutmix_i64: bad args");
  for (int64_t i = 0; i < (int64_t)B * HW; ++i) out[i] = box[i] == 1.f ? b[i] : a[i];
  return SVL_OK;
}
int svl_count_valid_i64(const int64_t* map, int64_t n, int64_t* count, svl_stream_t s) {
  (void)s;
  CHECK(map && count && n > 0, "svl_count_valid_i64: bad args");
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) c += map[i] != 255;
  *count += c;
  return SVL_OK;
}

/* one "block" per image */
int64_t svl_ce_num_blocks(int B, int N, int64_t HW) { return (B <= 0 || N <= 0 || HW <= 0 || N > 256) ? -1 : B; }

int svl_ce_fused_f32(const svl_ce_desc* d, svl_stream_t s) {
  (void)s;
  CHECK(d && d->logits && d->target && d->partials, "svl_ce_fused_f32: null args");
  CHECK(d->B > 0 && d->N > 0 && d->HW > 0, "svl_ce_fused_f32: bad sizes");
  CHECK((d->conf == NULL) == (d->ign == NULL), "svl_ce_fused_f32: conf and ign go together");
  CHECK(d->dlogits == NULL || d->gscale != NULL, "svl_ce_fused_f32: gscale required with dlogits");
  const int N = d->N;
  const int64_t HW = d->HW;
  for (int b = 0; b < d->B; ++b) {
    double st = 0, sm = 0, sc = 0, nv = 0;
    for (int64_t p = 0; p < HW; ++p) {
      const float* x = d->logits + (int64_t)b * N * HW + p;
      const int64_t o = (int64_t)b * HW + p;
      float m = -INFINITY;
      for (int c = 0; c < N; ++c) m = fmaxf(m, x[(int64_t)c * HW]);
      float sum = 0.f;
      for (int c = 0; c < N; ++c) sum += expf(x[(int64_t)c * HW] - m);
      const float lse = m + logf(sum);
      const int64_t t = d->target[o];
      const int t_ok = !(d->use_ignore_t && t == 255);
      float w = 1.f;
      int valid = t_ok;
      if (d->conf) {
        const int v = d->ign[o] != 255;
        const float cf = d->conf[o];
        w = d->all_pixels ? 1.f : ((cf >= d->conf_thresh && v) ? 1.f : 0.f);
        if (d->img_weight) w *= d->img_weight[b];       /* 'pixelratio' (train_utils.py:39-42) */
        valid = v;
        if (v) sc += cf;
      }
      nv += valid;
      int ti = -1, mi = -1;
      if (t_ok) { ti = (int)t; st += w * (lse - x[(int64_t)ti * HW]); }
      if (d->mc_target && d->mc_target[o] != 255) { mi = (int)d->mc_target[o]; sm += lse - x[(int64_t)mi * HW]; }
      if (d->dlogits) {
        const float gt = t_ok ? d->gscale[0] * w : 0.f, gm = mi >= 0 ? d->gscale[1] : 0.f;
        float* dl = d->dlogits + (int64_t)b * N * HW + p;
        for (int c = 0; c < N; ++c) {
          float g = (gt + gm) * (expf(x[(int64_t)c * HW] - m) / sum);
          if (c == ti) g -= gt;
          if (c == mi) g -= gm;
          dl[(int64_t)c * HW] = g;
        }
      }
    }
    float* pp = d->partials + 4 * b;
    pp[0] = (float)st; pp[1] = (float)sm; pp[2] = (float)sc; pp[3] = (float)nv;
  }
  return SVL_OK;
}
/* ---- the same two pixel-loss entry points on logits at the head's resolution (include/semivl_hip.h, round 5): the plain
 * restatement of  F.interpolate(logits, (H, W), mode='bilinear', align_corners)  (vlg_head.py:247, builder.py:93-97; ATen
 * upsample_bilinear2d: area_pixel_compute_source_index + guard_index_and_lambda)  followed by semivl.py:232,252 /
 * semivl.py:267-323, and of autograd's scatter of the resized gradient back onto the [h, w] grid.  No tiling: one partial
 * per image; every upsampling geometry of ratio <= 4.5 is taken (the HIP kernels additionally bound a tile's region). */
static void src_index_a(int dst, float scale, int in, int align, int* i0, int* i1, float* l0, float* l1) {
  float s = align ? scale * dst : scale * (dst + 0.5f) - 0.5f;
  if (!align && s < 0.f) s = 0.f;
  *i0 = (int)s < in - 1 ? (int)s : in - 1;
  *i1 = *i0 + 1 < in - 1 ? *i0 + 1 : in - 1;
  *l1 = fminf(fmaxf(s - *i0, 0.f), 1.f);
  *l0 = 1.f - *l1;
}
static float up_scale(int in, int out, int align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
static int up_ok(int N, int h, int w, int H, int W, int align) {
  return N > 0 && N <= 160 && h >= 2 && w >= 2 && H >= h && W >= w && up_scale(h, H, align) >= 2.f / 9.f &&
         up_scale(w, W, align) >= 2.f / 9.f;
}
int64_t svl_ce_up_num_blocks(int B, int N, int h, int w, int H, int W, int align) {
  return (B > 0 && up_ok(N, h, w, H, W, align)) ? B : -1;
}
/* the N resized logits of pixel (oy, ox) of image `x` [N, h, w] */
static void up_pixel(const float* x, int N, int h, int w, int y0, int y1, float ly0, float ly1, int x0, int x1, float lx0,
                     float lx1, float* v) {
  for (int c = 0; c < N; ++c) {
    const float* t = x + (int64_t)c * h * w;
    v[c] = ly0 * (lx0 * t[y0 * w + x0] + lx1 * t[y0 * w + x1]) + ly1 * (lx0 * t[y1 * w + x0] + lx1 * t[y1 * w + x1]);
  }
}
int svl_softmax_max_up_f32(const float* logits, int B, int N, int h, int w, int H, int W, int align, float* conf,
                           int64_t* label, svl_stream_t s) {
  (void)s;
  CHECK(logits && conf && label && B > 0, "svl_softmax_max_up_f32: bad args");
  CHECK(up_ok(N, h, w, H, W, align), "svl_softmax_max_up_f32: unsupported geometry");
  const float sh = up_scale(h, H, align), sw = up_scale(w, W, align);
  float v[160];
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < H; ++oy)
      for (int ox = 0; ox < W; ++ox) {
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index_a(oy, sh, h, align, &y0, &y1, &ly0, &ly1);
        src_index_a(ox, sw, w, align, &x0, &x1, &lx0, &lx1);
        up_pixel(logits + (int64_t)b * N * h * w, N, h, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1, v);
        float m = v[0];
        int idx = 0;
        for (int c = 1; c < N; ++c)
          if (v[c] > m) { m = v[c]; idx = c; }           /* first maximum wins (torch.max) */
        float sum = 0.f;
        for (int c = 0; c < N; ++c) sum += expf(v[c] - m);
        conf[((int64_t)b * H + oy) * W + ox] = 1.f / sum;
        label[((int64_t)b * H + oy) * W + ox] = idx;
      }
  return SVL_OK;
}
int svl_ce_up_fused_f32(const svl_ce_up_desc* d, svl_stream_t s) {
  (void)s;
  CHECK(d && d->logits && d->target && d->partials, "svl_ce_up_fused_f32: null args");
  CHECK(d->B > 0 && up_ok(d->N, d->h, d->w, d->H, d->W, d->align_corners), "svl_ce_up_fused_f32: unsupported geometry");
  CHECK((d->conf == NULL) == (d->ign == NULL), "svl_ce_up_fused_f32: conf and ign go together");
  CHECK(d->dlogits == NULL || d->gscale != NULL, "svl_ce_up_fused_f32: gscale required with dlogits");
  const int N = d->N, h = d->h, w = d->w, H = d->H, W = d->W, al = d->align_corners;
  const float sh = up_scale(h, H, al), sw = up_scale(w, W, al);
  float v[160];
  double* acc = d->dlogits ? (double*)malloc(sizeof(double) * (size_t)N * h * w) : NULL;
  CHECK(!d->dlogits || acc, "svl_ce_up_fused_f32: out of memory");
  for (int b = 0; b < d->B; ++b) {
    double st = 0, sm = 0, sc = 0, nv = 0;
    const float* x = d->logits + (int64_t)b * N * h * w;
    if (acc) memset(acc, 0, sizeof(double) * (size_t)N * h * w);
    for (int oy = 0; oy < H; ++oy)
      for (int ox = 0; ox < W; ++ox) {
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index_a(oy, sh, h, al, &y0, &y1, &ly0, &ly1);
        src_index_a(ox, sw, w, al, &x0, &x1, &lx0, &lx1);
        up_pixel(x, N, h, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1, v);
        const int64_t o = ((int64_t)b * H + oy) * W + ox;
        float m = -INFINITY;
        for (int c = 0; c < N; ++c) m = fmaxf(m, v[c]);
        float sum = 0.f;
        for (int c = 0; c < N; ++c) sum += expf(v[c] - m);
        const float lse = m + logf(sum);
        const int64_t t = d->target[o];
        const int t_ok = !(d->use_ignore_t && t == 255);
        float wt = 1.f;
        int valid = t_ok;
        if (d->conf) {
          const int vv = d->ign[o] != 255;
          const float cf = d->conf[o];
          wt = d->all_pixels ? 1.f : ((cf >= d->conf_thresh && vv) ? 1.f : 0.f);
          if (d->img_weight) wt *= d->img_weight[b];
          valid = vv;
          if (vv) sc += cf;
        }
        nv += valid;
        int ti = -1, mi = -1;
        if (t_ok) { ti = (int)t; st += wt * (lse - v[ti]); }
        if (d->mc_target && d->mc_target[o] != 255) { mi = (int)d->mc_target[o]; sm += lse - v[mi]; }
        if (acc) {
          const float gt = t_ok ? d->gscale[0] * wt : 0.f, gm = mi >= 0 ? d->gscale[1] : 0.f;
          for (int c = 0; c < N; ++c) {
            double g = (double)(gt + gm) * (double)(expf(v[c] - m) / sum);
            if (c == ti) g -= gt;
            if (c == mi) g -= gm;
            double* a = acc + (int64_t)c * h * w;           /* F.interpolate's backward: the four taps of the pixel */
            a[y0 * w + x0] += (double)(ly0 * lx0) * g;
            a[y0 * w + x1] += (double)(ly0 * lx1) * g;
            a[y1 * w + x0] += (double)(ly1 * lx0) * g;
            a[y1 * w + x1] += (double)(ly1 * lx1) * g;
          }
        }
      }
    if (acc) {
      float* dl = d->dlogits + (int64_t)b * N * h * w;
      for (int64_t i = 0; i < (int64_t)N * h * w; ++i) dl[i] = (float)acc[i];
    }
    float* pp = d->partials + 4 * b;
    pp[0] = (float)st; pp[1] = (float)sm; pp[2] = (float)sc; pp[3] = (float)nv;
  }
  free(acc);
  return SVL_OK;
}

int svl_ce_finalize(const float* partials, int64_t nblocks, double* sums, svl_stream_t s) {
  (void)s;
  CHECK(partials && sums && nblocks > 0, "svl_ce_finalize: bad args");
  for (int k = 0; k < 4; ++k) {
    sums[k] = 0;
    for (int64_t i = 0; i < nblocks; ++i) sums[k] += partials[4 * i + k];
  }
  return SVL_OK;
}

int svl_semivl_gscale(const int64_t* counts, double numel_u, float lam, const double* f, const int64_t* mc, float* g,
                      svl_stream_t s) {
  (void)s;
  CHECK(counts && g && numel_u > 0, "svl_semivl_gscale: bad args");
  const double f1 = f ? f[0] : 1.0, f2 = f ? f[1] : 1.0, f3 = f ? f[2] : 1.0;
  const double n1 = mc ? (double)mc[0] : numel_u, n2 = mc ? (double)mc[1] : numel_u, n3 = mc ? (double)mc[2] : numel_u;
  g[0] = (float)(0.5 / counts[0]);          g[1] = 0.f;
  g[2] = (float)(0.125 * f1 / counts[1]);   g[3] = (float)(0.25 * lam / n1);
  g[4] = (float)(0.125 * f2 / counts[2]);   g[5] = (float)(0.25 * lam / n2);
  g[6] = (float)(0.25 * f3 / counts[3]);    g[7] = (float)(0.5 * lam / n3);
  return SVL_OK;
}
int svl_semivl_loss(const double* sums, double numel_u, float lam, const double* f, const int64_t* mc, float* out,
                    svl_stream_t s) {
  (void)s;
  CHECK(sums && out && numel_u > 0, "svl_semivl_loss: bad args");
  const double f1 = f ? f[0] : 1.0, f2 = f ? f[1] : 1.0, f3 = f ? f[2] : 1.0;
  const double n1 = mc ? (double)mc[0] : numel_u, n2 = mc ? (double)mc[1] : numel_u, n3 = mc ? (double)mc[2] : numel_u;
  const float lx = (float)(sums[0] / sums[3]), l1 = (float)(sums[4] * f1 / sums[7]);
  const float l2 = (float)(sums[8] * f2 / sums[11]), lf = (float)(sums[12] * f3 / sums[15]);
  const float m1 = (float)(sums[5] / n1), m2 = (float)(sums[9] / n2), mf = (float)(sums[13] / n3);   /* semivl.py:52-58 */
  float loss = (lx + l1 * 0.25f + l2 * 0.25f + lf * 0.5f) / 2.0f;   /* semivl.py:312-316 */
  loss = loss + m1 * 0.25f * lam;                                    /* semivl.py:317-323 */
  loss = loss + m2 * 0.25f * lam;
  loss = loss + mf * 0.5f * lam;
  out[0] = loss; out[1] = lx; out[2] = l1; out[3] = l2; out[4] = lf; out[5] = m1; out[6] = m2; out[7] = mf;
  return SVL_OK;
}
int64_t svl_conf_avg_ws_doubles(int B) { return B > 0 ? 2 * (int64_t)B : 0; }
int svl_conf_avg_factor(const float* conf, const int64_t* ign, int B, int64_t HW, double* factor, double* ws, svl_stream_t s) {
  (void)s; (void)ws;
  CHECK(conf && ign && factor && B > 0 && HW > 0, "svl_conf_avg_factor: bad args");
  double tot = 0;
  for (int b = 0; b < B; ++b) {
    double sc = 0, nv = 0;
    for (int64_t p = 0; p < HW; ++p)
      if (ign[(int64_t)b * HW + p] != 255) { sc += conf[(int64_t)b * HW + p]; nv += 1; }
    tot += sc / nv;
  }
  *factor = tot;
  return SVL_OK;
}

int svl_conf_ratio_f32(const float* conf, const int64_t* ign, int B, int64_t HW, float thresh, float* ratio, double* ws,
                       svl_stream_t s) {   /* train_utils.py:39-40 */
  (void)s; (void)ws;
  CHECK(conf && ign && ratio && B > 0 && HW > 0 && thresh >= 0.f, "svl_conf_ratio_f32: bad args");
  for (int b = 0; b < B; ++b) {
    int64_t hi = 0, nv = 0;
    for (int64_t p = 0; p < HW; ++p)
      if (ign[(int64_t)b * HW + p] != 255) { hi += conf[(int64_t)b * HW + p] >= thresh; nv += 1; }
    ratio[b] = (float)hi / (float)nv;
  }
  return SVL_OK;
}

static void src_index(int dst, float scale, int in, int* i0, int* i1, float* l0, float* l1) { /* align_corners=False */
  float s = scale * (dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  *i0 = (int)s < in - 1 ? (int)s : in - 1;
  *i1 = *i0 + 1 < in - 1 ? *i0 + 1 : in - 1;
  *l1 = fminf(fmaxf(s - *i0, 0.f), 1.f);
  *l0 = 1.f - *l1;
}
int svl_maskclip_labels(const float* dense, int B, int N, int h, int w, int H, int W, float scale, float thresh,
                        const int64_t* ign, int64_t* out, svl_stream_t s) {
  (void)s;
  CHECK(dense && out && B > 0 && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "svl_maskclip_labels: bad args");
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < H; ++oy)
      for (int ox = 0; ox < W; ++ox) {
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(oy, sh, h, &y0, &y1, &ly0, &ly1);
        src_index(ox, sw, w, &x0, &x1, &lx0, &lx1);
        float m = -INFINITY, sum = 0.f;
        int idx = 0;
        for (int c = 0; c < N; ++c) {
          const float* pl = dense + ((int64_t)b * N + c) * h * w;
          const float v = ly0 * (lx0 * pl[y0 * w + x0] + lx1 * pl[y0 * w + x1]) +
                          ly1 * (lx0 * pl[y1 * w + x0] + lx1 * pl[y1 * w + x1]);
          const float x = scale * v;
          if (x > m) { sum = sum * expf(m - x) + 1.f; m = x; idx = c; } else { sum += expf(x - m); }
        }
        const int64_t i = ((int64_t)b * H + oy) * W + ox;
        int64_t lab = (1.f / sum < thresh) ? 255 : idx;
        if (ign && ign[i] == 255) lab = 255;
        out[i] = lab;
      }
  return SVL_OK;
}
int svl_concept_max_f32(const float* pred, int B, int NC, int64_t HW, const int* off, int N, float* out, svl_stream_t s) {
  (void)s;
  CHECK(pred && off && out && B > 0 && NC > 0 && HW > 0 && N > 0, "svl_concept_max_f32: bad args");
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < N; ++c)
      for (int64_t p = 0; p < HW; ++p) {
        float m = -INFINITY;
        for (int k = off[c]; k < off[c + 1]; ++k) m = fmaxf(m, pred[((int64_t)b * NC + k) * HW + p]);
        out[((int64_t)b * N + c) * HW + p] = m;
      }
  return SVL_OK;
}
int svl_iou_hist_i64(const int64_t* pred, const int64_t* tgt, int64_t n, int K, int ignore, int64_t* hist, svl_stream_t s) {
  (void)s;
  CHECK(pred && tgt && hist && n > 0 && K > 0, "svl_iou_hist_i64: bad args");
  for (int64_t i = 0; i < n; ++i) {
    const int64_t t = tgt[i], o = t == ignore ? ignore : pred[i];
    if (o >= 0 && o < K) { hist[K + o] += 1; if (o == t) hist[o] += 1; }
    if (t >= 0 && t < K) hist[2 * K + t] += 1;
  }
  return SVL_OK;
}

int svl_adamw_step(float* p, const float* g, float* m, float* v, const int64_t* seg_off, const float* seg_lr,
                   const float* seg_wd, int nseg, int64_t total, float beta1, float beta2, float eps, int step,
                   float gscale, float* ema, float ema_decay, svl_stream_t s) {
  (void)s;
  CHECK(p && g && m && v && seg_off && seg_lr && seg_wd && nseg > 0 && total > 0 && step >= 1, "svl_adamw_step: bad args");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  int seg = 0;
  for (int64_t i = 0; i < total; ++i) {
    while (seg + 1 < nseg && seg_off[seg + 1] <= i) ++seg;
    const float lr = seg_lr[seg], wd = seg_wd[seg], gr = g[i] * gscale;
    float pw = p[i] * (1.f - lr * wd);                       /* decoupled weight decay */
    const float mm = m[i] + (gr - m[i]) * (1.f - beta1);     /* exp_avg.lerp_(grad, 1 - beta1) */
    const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
    pw -= (lr / bc1) * (mm / (sqrtf(vv) / bc2s + eps));
    p[i] = pw; m[i] = mm; v[i] = vv;
    if (ema) ema[i] = ema_decay * ema[i] + (1.f - ema_decay) * pw;
  }
  return SVL_OK;
}
