/*
 * ctd_oracle.c -- CPU restatement of the reference's disparity hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load the library built from
 * this file (oracle/libctd_oracle.so).  The product path
 * (connecting_the_dots_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below bit-for-bit (native ops) or to the stated tolerance (ATen-composed ops)
 * against tests/golden/ (.npz files), which tests/golden/make_golden.py generated in
 * the build container by running the reference itself: the unmodified
 * torchext/ext/ext_cpu.cpp compiled by oracle/build_ref.py, and
 * model/networks.py imported through oracle/ref_python.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (no -march=native: FMA contraction would break bit-parity with the
 * reference CPU build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define T float
#define SFX f32
#define SQRT sqrtf
#define FABS fabsf
#include "ctd_oracle_impl.h"
#undef T
#undef SFX
#undef SQRT
#undef FABS

#define T double
#define SFX f64
#define SQRT sqrt
#define FABS fabs
#include "ctd_oracle_impl.h"
#undef T
#undef SFX
#undef SQRT
#undef FABS

/* ------------------------------------------------------------------------- *
 * CrossCheckFunctor::operator()   torchext/ext/ext.h:49-66   (host ext_cpu.cpp:39-56)
 * in0 int64 [n0] (indices into in1), in1 int64 [n1] (indices into in0) -> out u8 [n0]:
 * 1 where the match is mutual.  `int idx1 = in0[idx0]` truncates to int (ext.h:61).
 * The reference reads in1[idx1] without an upper bound check (undefined behaviour for
 * idx1 >= n1); restated as "not mutual".
 * ------------------------------------------------------------------------- */
int ctd_oracle_crosscheck(const int64_t* in0, const int64_t* in1, long nelem0, long nelem1, uint8_t* out) {
  for (long idx0 = 0; idx0 < nelem0; ++idx0) {
    int idx1 = (int)in0[idx0];
    out[idx0] = idx1 >= 0 && idx1 < nelem1 && in1[idx1] >= 0 && idx0 == in1[idx1];
  }
  return 0;
}

/* reflect index without repeating the edge (torch.nn.ReflectionPad2d) */
static inline int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i > n - 1) i = 2 * (n - 1) - i;
  return i;
}

/* ------------------------------------------------------------------------- *
 * LCN.tforward   model/networks.py:507-533   (f32, [N,1,H,W])
 *   boxs  = ones-conv over ReflectionPad2d(r)          :513-518, :524
 *   avgs  = boxs / (2r+1)^2                            :526
 *   stds  = sqrt(box(x^2)/(2r+1)^2 - avgs^2 + 1e-6)+eps:528-531
 *   out   = (x - avgs) / stds,  stds                   :533
 * The reference's box sums come out of ATen's conv2d whose summation order is
 * unspecified; this restatement sums separably (dx ascending inside, dy
 * ascending outside) in double and rounds once to float, which is within
 * half an ulp of the exact sum, then follows the reference's f32 elementwise
 * order.  Parity against the reference is therefore by tolerance (tests).
 * ------------------------------------------------------------------------- */
int ctd_oracle_lcn_f32(const float* x, float* y, float* stds, int n, int height,
                       int width, int radius, float eps) {
  const int ks = 2 * radius + 1;
  const float cnt = (float)(ks * ks);
  if (radius < 0 || radius >= height || radius >= width) return 1;
  double* r1 = (double*)malloc(sizeof(double) * (size_t)height * width);
  double* r2 = (double*)malloc(sizeof(double) * (size_t)height * width);
  if (!r1 || !r2) { free(r1); free(r2); return 2; }
  for (int b = 0; b < n; ++b) {
    const float* xb = x + (size_t)b * height * width;
    for (int h = 0; h < height; ++h)
      for (int w = 0; w < width; ++w) {
        double s1 = 0, s2 = 0;
        for (int dx = -radius; dx <= radius; ++dx) {
          float v = xb[(size_t)h * width + reflect_idx(w + dx, width)];
          float v2 = v * v;                     /* data**2 is an f32 tensor, :528 */
          s1 += (double)v;
          s2 += (double)v2;
        }
        r1[(size_t)h * width + w] = s1;
        r2[(size_t)h * width + w] = s2;
      }
    for (int h = 0; h < height; ++h)
      for (int w = 0; w < width; ++w) {
        double s1 = 0, s2 = 0;
        for (int dy = -radius; dy <= radius; ++dy) {
          int hh = reflect_idx(h + dy, height);
          s1 += r1[(size_t)hh * width + w];
          s2 += r2[(size_t)hh * width + w];
        }
        float boxs = (float)s1, boxs_2n = (float)s2;
        float avgs = boxs / cnt;
        float var = boxs_2n / cnt - avgs * avgs + 1e-6f;
        float sd = sqrtf(var) + eps;
        size_t o = ((size_t)b * height + h) * width + w;
        y[o] = (xb[(size_t)h * width + w] - avgs) / sd;
        stds[o] = sd;
      }
  }
  free(r1);
  free(r2);
  return 0;
}

/* ------------------------------------------------------------------------- *
 * data/lcn/lcn.pyx:16-58  (Cython data-generation variant; f32 image [H,W])
 * two-pass window mean / std over (2ks+1)^2, zero border of width ks,
 * out = (x-mean)/(std+eps); returns the RAW std image.
 * ------------------------------------------------------------------------- */
int ctd_oracle_lcn_datagen_f32(const float* img, float* out, float* out_std,
                               int height, int width, int ks, float eps) {
  memset(out, 0, sizeof(float) * (size_t)height * width);
  memset(out_std, 0, sizeof(float) * (size_t)height * width);
  const float num = (float)((ks * 2 + 1) * (ks * 2 + 1));   /* lcn.pyx:26 */
  for (int y = ks; y < height - ks; ++y)
    for (int x = ks; x < width - ks; ++x) {
      float mean = 0;
      for (int i = -ks; i <= ks; ++i)
        for (int j = -ks; j <= ks; ++j)
          mean += img[(size_t)(y + i) * width + x + j];
      mean = mean / num;
      float std = 0;
      for (int i = -ks; i <= ks; ++i)
        for (int j = -ks; j <= ks; ++j) {
          float t = img[(size_t)(y + i) * width + x + j] - mean;
          std += t * t;
        }
      std = sqrtf(std / num);
      out[(size_t)y * width + x] = (img[(size_t)y * width + x] - mean) / (std + eps);
      out_std[(size_t)y * width + x] = std;
    }
  return 0;
}

/* ------------------------------------------------------------------------- *
 * DispToDepth.tforward   model/networks.py:313-321
 *   disp = relu(disp) + 1e-12 ;  depth = bf / disp
 * "python_float / tensor" dispatches to Tensor.__rtruediv__ = reciprocal()*bf,
 * i.e. two roundings; restated the same way.
 * ------------------------------------------------------------------------- */
int ctd_oracle_disp_to_depth_f32(const float* disp, float* depth, long n, float bf) {
  for (long i = 0; i < n; ++i) {
    float d = disp[i] > 0.f ? disp[i] : 0.f;
    d = d + 1e-12f;
    depth[i] = (1.0f / d) * bf;
  }
  return 0;
}

/* ------------------------------------------------------------------------- *
 * RenderProjectorFunctor<float>::operator()   renderer/render/render.h:251-364
 * (ray / mesh intersection geometry.h:201-258, camera render.h:12-87, Phong shader
 * geometry.h:262-292, bilinear pattern fetch render.h:228-249), f32 only as the
 * reference instantiates it (render_cpu.cpp:22).
 *   verts, colors [n_verts][3]; faces int [n_faces][3];
 *   cam / proj = { fx, fy, px, py, R[9] row-major, t[3] } + width / height;
 *   shader = { ka, kd, ks, alpha }; pattern [proj_h][proj_w][3]
 *   -> depth [H][W] (-1 = no hit), color [H][W][3] (projected pattern, distance decay),
 *      normal [H][W][3] (Phong-shaded vertex colours: the "ambient" image; left untouched
 *      where the camera ray hits nothing, as in the reference)
 * ------------------------------------------------------------------------- */
#include <float.h>

typedef struct {
  float fx, fy, px, py, R[9], t[3], C[3];
  int width, height;
} ocam_t;

static void ocam_init(ocam_t* c, const float* p, int w, int h) {
  c->fx = p[0]; c->fy = p[1]; c->px = p[2]; c->py = p[3];
  for (int i = 0; i < 9; ++i) c->R[i] = p[4 + i];
  for (int i = 0; i < 3; ++i) c->t[i] = p[13 + i];
  const float* R = c->R; const float* t = c->t;
  c->C[0] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);     /* render.h:29-31 */
  c->C[1] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
  c->C[2] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
  c->width = w; c->height = h;
}

static inline float odot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void ocross3(const float* u, const float* v, float* o) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}
static inline float onorm3(const float* u) { return sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]); }
static inline void onormalize3(const float* u, float* v) {
  const float n = onorm3(u);
  v[0] = u[0] / n; v[1] = u[1] / n; v[2] = u[2] / n;
}

/* geometry.h:201-233 (Moeller-Trumbore; returns barycentrics re-ordered as the reference does) */
static int oray_tri(const float* orig, const float* dir, const float* v0, const float* v1, const float* v2, float* t,
                    float* u, float* v) {
  const float eps = 1e-6f;
  float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
  float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
  float pvec[3];
  ocross3(dir, e2, pvec);
  const float det = odot3(e1, pvec);
  if (fabsf(det) < eps) return 0;
  const float inv_det = 1 / det;
  float tvec[3] = {orig[0] - v0[0], orig[1] - v0[1], orig[2] - v0[2]};
  *u = odot3(tvec, pvec) * inv_det;
  if (*u < 0 || *u > 1) return 0;
  float qvec[3];
  ocross3(tvec, e1, qvec);
  *v = odot3(dir, qvec) * inv_det;
  if (*v < 0 || (*u + *v) > 1) return 0;
  *t = odot3(e2, qvec) * inv_det;
  const float w = 1 - *u - *v;
  *v = *u;
  *u = w;
  return 1;
}

/* geometry.h:235-258 */
static int oray_mesh(const float* orig, const float* dir, const int* faces, int n_faces, const float* verts,
                     int* face_idx, float* t, float* u, float* v) {
  *t = FLT_MAX;
  int valid = 0;
  for (int f = 0; f < n_faces; ++f) {
    float ft, fu, fv;
    if (oray_tri(orig, dir, verts + faces[f * 3 + 0] * 3, verts + faces[f * 3 + 1] * 3, verts + faces[f * 3 + 2] * 3,
                 &ft, &fu, &fv) && ft < *t) {
      *face_idx = f; *t = ft; *u = fu; *v = fv;
      valid = 1;
    }
  }
  return valid;
}

static inline float omax(float a, float b) { return a < b ? b : a; }     /* std::max(a, b) */
static inline float omin(float a, float b) { return b < a ? b : a; }     /* std::min(a, b) */

int ctd_oracle_render_mesh_proj_f32(const float* verts, const float* colors, int n_verts, const int* faces,
                                    int n_faces, const float* cam_p, int cam_w, int cam_h, const float* proj_p,
                                    int proj_w, int proj_h, const float* shader, const float* pattern, float d_alpha,
                                    float d_beta, float* depth, float* color, float* normal, int nthreads) {
  ocam_t cam, proj;
  ocam_init(&cam, cam_p, cam_w, cam_h);
  ocam_init(&proj, proj_p, proj_w, proj_h);
  const float ka = shader[0], kd = shader[1], ks = shader[2], alpha = shader[3];
  (void)n_verts; (void)nthreads;
  int idx;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (idx = 0; idx < cam_w * cam_h; ++idx) {
    const int h = idx / cam.width, w = idx % cam.width;
    const float* orig = cam.C;
    float dir[3];
    {                                                           /* Camera::to_ray, render.h:52-60 */
      const float u0 = (w - cam.px) / cam.fx, u1 = (h - cam.py) / cam.fy;
      dir[0] = cam.R[0] * u0 + cam.R[3] * u1 + cam.R[6];
      dir[1] = cam.R[1] * u0 + cam.R[4] * u1 + cam.R[7];
      dir[2] = cam.R[2] * u0 + cam.R[5] * u1 + cam.R[8];
    }
    int face_idx = 0;
    float t, tu, tv;
    int valid = oray_mesh(orig, dir, faces, n_faces, verts, &face_idx, &t, &tu, &tv);
    if (depth) depth[idx] = valid ? t : -1;
    color[idx * 3 + 0] = 0; color[idx * 3 + 1] = 0; color[idx * 3 + 2] = 0;
    if (!valid) continue;
    if (normal) {                                               /* render.h:283-312 */
      const int* face = faces + face_idx * 3;
      const float tw = 1 - tu - tv;
      const float *a = verts + face[0] * 3, *b = verts + face[1] * 3, *c = verts + face[2] * 3;
      float e1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, e2[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
      float norm[3];
      ocross3(e1, e2, norm);
      onormalize3(norm, norm);
      if (odot3(norm, dir) > 0) { norm[0] = norm[0] * -1.f; norm[1] = norm[1] * -1.f; norm[2] = norm[2] * -1.f; }
      float col[3] = {0.f, 0.f, 0.f};
      const float bary[3] = {tu, tv, tw};
      for (int k = 0; k < 3; ++k) {                             /* vec_add(1.f, color, lam, colors + face[k]*3, color) */
        const float* cv = colors + face[k] * 3;
        col[0] = 1.f * col[0] + bary[k] * cv[0];
        col[1] = 1.f * col[1] + bary[k] * cv[1];
        col[2] = 1.f * col[2] + bary[k] * cv[2];
      }
      float sp[3] = {1.f * orig[0] + t * dir[0], 1.f * orig[1] + t * dir[1], 1.f * orig[2] + t * dir[2]};
      /* reflectance_phong(orig, sp, lp = orig, n), geometry.h:277-292 */
      float l[3] = {orig[0] - sp[0], orig[1] - sp[1], orig[2] - sp[2]};
      onormalize3(l, l);
      const float two_ln = 2 * odot3(l, norm);
      float r[3] = {two_ln * norm[0] + -1.f * l[0], two_ln * norm[1] + -1.f * l[1], two_ln * norm[2] + -1.f * l[2]};
      onormalize3(r, r);
      float v[3] = {orig[0] - sp[0], orig[1] - sp[1], orig[2] - sp[2]};
      onormalize3(v, v);
      const float refl = ka + kd * odot3(l, norm) + ks * powf(odot3(r, v), alpha);
      for (int k = 0; k < 3; ++k) normal[idx * 3 + k] = omin(1.f, omax(0.f, refl * col[k]));
    }
    float pt[3] = {dir[0] * t, dir[1] * t, dir[2] * t};
    pt[0] = orig[0] + pt[0]; pt[1] = orig[1] + pt[1]; pt[2] = orig[2] + pt[2];
    const float* porig = proj.C;
    float pdir[3] = {pt[0] - porig[0], pt[1] - porig[1], pt[2] - porig[2]};
    { const float z = pdir[2]; pdir[0] = pdir[0] / z; pdir[1] = pdir[1] / z; pdir[2] = pdir[2] / z; }
    int p_face = 0;
    float p_t, p_tu, p_tv;
    valid = oray_mesh(porig, pdir, faces, n_faces, verts, &p_face, &p_t, &p_tu, &p_tv);
    float p_pt[3] = {pdir[0] * p_t, pdir[1] * p_t, pdir[2] * p_t};
    p_pt[0] = porig[0] + p_pt[0]; p_pt[1] = porig[1] + p_pt[1]; p_pt[2] = porig[2] + p_pt[2];
    float diff[3] = {p_pt[0] - pt[0], p_pt[1] - pt[1], p_pt[2] - pt[2]};
    if (!valid || onorm3(diff) > 1e-5) continue;                /* float compared with a double literal, render.h:338 */
    /* Camera::to_2d, render.h:62-71 */
    float y[3];
    y[0] = proj.R[0] * p_pt[0] + proj.R[1] * p_pt[1] + proj.R[2] * p_pt[2] + proj.t[0];
    y[1] = proj.R[3] * p_pt[0] + proj.R[4] * p_pt[1] + proj.R[5] * p_pt[2] + proj.t[1];
    y[2] = proj.R[6] * p_pt[0] + proj.R[7] * p_pt[1] + proj.R[8] * p_pt[2] + proj.t[2];
    float u = proj.fx * y[0] + proj.px * y[2];
    float v = proj.fy * y[1] + proj.py * y[2];
    const float d = y[2];
    u /= d;
    v /= d;
    if (u >= 0 && v >= 0 && u < proj.width && v < proj.height) {
      /* interpolate_linear, render.h:228-249 */
      int x1 = (int)u, y1 = (int)v;
      int x2 = x1 + 1, y2 = y1 + 1;
      const float denom = (float)((x2 - x1) * (y2 - y1));
      const float t11 = (x2 - u) * (y2 - v);
      const float t21 = (u - x1) * (y2 - v);
      const float t12 = (x2 - u) * (v - y1);
      const float t22 = (u - x1) * (v - y1);
      x1 = x1 < 0 ? 0 : x1; x1 = x1 > proj.width - 1 ? proj.width - 1 : x1;
      x2 = x2 < 0 ? 0 : x2; x2 = x2 > proj.width - 1 ? proj.width - 1 : x2;
      y1 = y1 < 0 ? 0 : y1; y1 = y1 > proj.height - 1 ? proj.height - 1 : y1;
      y2 = y2 < 0 ? 0 : y2; y2 = y2 > proj.height - 1 ? proj.height - 1 : y2;
      for (int k = 0; k < 3; ++k)
        color[idx * 3 + k] = (pattern[(y1 * proj.width + x1) * 3 + k] * t11 + pattern[(y2 * proj.width + x1) * 3 + k] * t12 +
                              pattern[(y1 * proj.width + x2) * 3 + k] * t21 + pattern[(y2 * proj.width + x2) * 3 + k] * t22) /
                             denom;
      float decay = d_alpha + d_beta * d;
      decay *= decay;
      decay = omax(decay, 1.f);
      for (int k = 0; k < 3; ++k) color[idx * 3 + k] = color[idx * 3 + k] / decay;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- *
 * RenderMeshFunctor<float>::operator()   renderer/render/render.h:150-223: camera rays only.
 *   normals [n_verts][3]: interpolated with the hit's barycentrics (not normalised), flipped towards the camera
 *   -> depth (-1 = no hit), color = clamp(phong * interpolated vertex colour), normal; colour and normal are
 *      zeroed where nothing is hit.  Any of the three outputs may be NULL.
 * ------------------------------------------------------------------------- */
int ctd_oracle_render_mesh_f32(const float* verts, const float* colors, const float* normals, int n_verts,
                               const int* faces, int n_faces, const float* cam_p, int cam_w, int cam_h,
                               const float* shader, float* depth, float* color, float* normal, int nthreads) {
  ocam_t cam;
  ocam_init(&cam, cam_p, cam_w, cam_h);
  const float ka = shader[0], kd = shader[1], ks = shader[2], alpha = shader[3];
  (void)n_verts; (void)nthreads;
  int idx;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (idx = 0; idx < cam_w * cam_h; ++idx) {
    const int h = idx / cam.width, w = idx % cam.width;
    const float* orig = cam.C;
    float dir[3];
    {
      const float u0 = (w - cam.px) / cam.fx, u1 = (h - cam.py) / cam.fy;
      dir[0] = cam.R[0] * u0 + cam.R[3] * u1 + cam.R[6];
      dir[1] = cam.R[1] * u0 + cam.R[4] * u1 + cam.R[7];
      dir[2] = cam.R[2] * u0 + cam.R[5] * u1 + cam.R[8];
    }
    int face_idx = 0;
    float t, tu, tv;
    const int valid = oray_mesh(orig, dir, faces, n_faces, verts, &face_idx, &t, &tu, &tv);
    if (depth) depth[idx] = valid ? t : -1;
    if (!valid) {
      for (int k = 0; k < 3; ++k) {
        if (color) color[idx * 3 + k] = 0;
        if (normal) normal[idx * 3 + k] = 0;
      }
      continue;
    }
    if (!normal && !color) continue;
    const int* face = faces + face_idx * 3;
    const float bary[3] = {tu, tv, 1 - tu - tv};
    float norm[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {                               /* render.h:190-193 */
      const float* nv = normals + face[k] * 3;
      norm[0] = 1.f * norm[0] + bary[k] * nv[0];
      norm[1] = 1.f * norm[1] + bary[k] * nv[1];
      norm[2] = 1.f * norm[2] + bary[k] * nv[2];
    }
    if (odot3(norm, dir) > 0) { norm[0] = norm[0] * -1.f; norm[1] = norm[1] * -1.f; norm[2] = norm[2] * -1.f; }
    if (normal) for (int k = 0; k < 3; ++k) normal[idx * 3 + k] = norm[k];
    if (color) {
      float col[3] = {0.f, 0.f, 0.f};
      for (int k = 0; k < 3; ++k) {
        const float* cv = colors + face[k] * 3;
        col[0] = 1.f * col[0] + bary[k] * cv[0];
        col[1] = 1.f * col[1] + bary[k] * cv[1];
        col[2] = 1.f * col[2] + bary[k] * cv[2];
      }
      float sp[3] = {1.f * orig[0] + t * dir[0], 1.f * orig[1] + t * dir[1], 1.f * orig[2] + t * dir[2]};
      float l[3] = {orig[0] - sp[0], orig[1] - sp[1], orig[2] - sp[2]};   /* reflectance_phong, geometry.h:277-292 */
      onormalize3(l, l);
      const float two_ln = 2 * odot3(l, norm);
      float r[3] = {two_ln * norm[0] + -1.f * l[0], two_ln * norm[1] + -1.f * l[1], two_ln * norm[2] + -1.f * l[2]};
      onormalize3(r, r);
      float v[3] = {orig[0] - sp[0], orig[1] - sp[1], orig[2] - sp[2]};
      onormalize3(v, v);
      const float refl = ka + kd * odot3(l, norm) + ks * powf(odot3(r, v), alpha);
      for (int k = 0; k < 3; ++k) color[idx * 3 + k] = omin(1.f, omax(0.f, refl * col[k]));
    }
  }
  return 0;
}
