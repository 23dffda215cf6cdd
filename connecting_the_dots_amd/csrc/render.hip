// render.hip -- synthetic structured-light rendering: the reference's brute-force ray caster
// with projector inpainting, RenderProjectorFunctor<float>::operator()
// (/root/reference/renderer/render/render.h:251-364; ray/mesh intersection geometry.h:201-258, camera
// render.h:12-87, Phong shader geometry.h:262-292, bilinear pattern fetch render.h:228-249; launched by
// render_gpu.cu through iterate_cuda, one thread per camera pixel).
//
// One thread per camera pixel; the mesh streams through LDS in tiles of gathered vertex triples that every lane
// reads at the same address (broadcast), once for the camera ray and once for the shadow ray from the
// projector.  Every expression keeps the reference's operation order and the library is built without FMA
// contraction, so depth, hit face and the projected pattern are bit-identical to the reference CPU build; the
// shaded ambient image is too whenever the specular weight ks is 0 (the data generator's setting,
// data/create_syn_data.py:155), otherwise it differs by powf's last bits.
#include <cfloat>

#include "ctd_internal.h"

namespace ctd {

struct CamDev {
  float fx, fy, px, py, R[9], t[3], C[3];
  int width, height;
};

static CamDev make_cam(const float* p, int w, int h) {
  CamDev c;
  c.fx = p[0]; c.fy = p[1]; c.px = p[2]; c.py = p[3];
  for (int i = 0; i < 9; ++i) c.R[i] = p[4 + i];
  for (int i = 0; i < 3; ++i) c.t[i] = p[13 + i];
  const float* R = c.R;
  const float* t = c.t;
  c.C[0] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);     // render.h:29-31
  c.C[1] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
  c.C[2] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
  c.width = w;
  c.height = h;
  return c;
}

__device__ inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline void cross3(const float* u, const float* v, float* o) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}
__device__ inline float norm3(const float* u) { return sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]); }
__device__ inline void normalize3(const float* u, float* v) {
  const float n = norm3(u);
  v[0] = u[0] / n; v[1] = u[1] / n; v[2] = u[2] / n;
}
__device__ inline float std_max(float a, float b) { return a < b ? b : a; }
__device__ inline float std_min(float a, float b) { return b < a ? b : a; }

// geometry.h:201-233
__device__ inline bool ray_tri(const float* orig, const float* dir, const float* v0, const float* v1, const float* v2,
                               float& t, float& u, float& v) {
  const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
  const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
  float pvec[3];
  cross3(dir, e2, pvec);
  const float det = dot3(e1, pvec);
  if (fabsf(det) < 1e-6f) return false;
  const float inv_det = 1 / det;
  const float tvec[3] = {orig[0] - v0[0], orig[1] - v0[1], orig[2] - v0[2]};
  u = dot3(tvec, pvec) * inv_det;
  if (u < 0 || u > 1) return false;
  float qvec[3];
  cross3(tvec, e1, qvec);
  v = dot3(dir, qvec) * inv_det;
  if (v < 0 || (u + v) > 1) return false;
  t = dot3(e2, qvec) * inv_det;
  const float w = 1 - u - v;
  v = u;
  u = w;
  return true;
}

constexpr int kFaceTile = 256;     // faces per LDS tile: 256 x 3 vertices x float4 = 12 KB

// nearest hit of one ray per thread against the whole mesh (geometry.h:235-258); all threads of the workgroup
// take part in the staging even if their ray is not live
__device__ inline bool ray_mesh(float (*tile)[4], const float* orig, const float* dir, bool live,
                                const float* __restrict__ verts, const int* __restrict__ faces, int n_faces,
                                int& face_idx, float& t, float& u, float& v) {
  t = FLT_MAX;
  bool valid = false;
  for (int base = 0; base < n_faces; base += kFaceTile) {
    const int n = min(kFaceTile, n_faces - base);
    __syncthreads();
    for (int i = threadIdx.x; i < n * 3; i += blockDim.x) {
      const int vi = faces[(long)base * 3 + i];
      tile[i][0] = verts[(long)vi * 3 + 0];
      tile[i][1] = verts[(long)vi * 3 + 1];
      tile[i][2] = verts[(long)vi * 3 + 2];
    }
    __syncthreads();
    if (!live) continue;
    for (int f = 0; f < n; ++f) {
      float ft, fu, fv;
      if (ray_tri(orig, dir, tile[3 * f], tile[3 * f + 1], tile[3 * f + 2], ft, fu, fv) && ft < t) {
        face_idx = base + f;
        t = ft;
        u = fu;
        v = fv;
        valid = true;
      }
    }
  }
  return valid;
}

// vec_add(1.f, acc, lam_k, attr + face[k] * 3, acc) for the three corners (render.h:199-203 / 302-306)
__device__ inline void bary_mix(const float* __restrict__ attr, const int* face, float tu, float tv, float tw, float* acc) {
  acc[0] = acc[1] = acc[2] = 0.f;
  const float bary[3] = {tu, tv, tw};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float* a = attr + (long)face[k] * 3;
    acc[0] = 1.f * acc[0] + bary[k] * a[0];
    acc[1] = 1.f * acc[1] + bary[k] * a[1];
    acc[2] = 1.f * acc[2] + bary[k] * a[2];
  }
}

// Shader::operator()(orig, sp, lp = orig, n) with sp = orig + t * dir: reflectance_phong, geometry.h:277-292
__device__ inline float phong_at_camera(const float* orig, const float* dir, float t, const float* nrm, float ka, float kd,
                                        float ks, float alpha) {
  const float sp[3] = {1.f * orig[0] + t * dir[0], 1.f * orig[1] + t * dir[1], 1.f * orig[2] + t * dir[2]};
  float l[3] = {orig[0] - sp[0], orig[1] - sp[1], orig[2] - sp[2]};       // light at the camera centre
  normalize3(l, l);
  const float two_ln = 2 * dot3(l, nrm);
  float r[3] = {two_ln * nrm[0] + -1.f * l[0], two_ln * nrm[1] + -1.f * l[1], two_ln * nrm[2] + -1.f * l[2]};
  normalize3(r, r);
  float vv[3] = {orig[0] - sp[0], orig[1] - sp[1], orig[2] - sp[2]};
  normalize3(vv, vv);
  return ka + kd * dot3(l, nrm) + ks * powf(dot3(r, vv), alpha);
}

__device__ inline void camera_ray(const CamDev& cam, int h, int w, float* dir) {   // Camera::to_ray, render.h:52-60
  const float u0 = (w - cam.px) / cam.fx, u1 = (h - cam.py) / cam.fy;
  dir[0] = cam.R[0] * u0 + cam.R[3] * u1 + cam.R[6];
  dir[1] = cam.R[1] * u0 + cam.R[4] * u1 + cam.R[7];
  dir[2] = cam.R[2] * u0 + cam.R[5] * u1 + cam.R[8];
}

__global__ __launch_bounds__(256) void render_proj_kernel(const float* __restrict__ verts, const float* __restrict__ colors,
                                                          const int* __restrict__ faces, int n_faces, CamDev cam,
                                                          CamDev proj, float ka, float kd, float ks, float alpha,
                                                          const float* __restrict__ pattern, float d_alpha, float d_beta,
                                                          float* __restrict__ depth, float* __restrict__ color,
                                                          float* __restrict__ normal) {
  __shared__ float tile[kFaceTile * 3][4];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_img = idx < cam.width * cam.height;
  const int h = idx / cam.width, w = idx % cam.width;
  const float orig[3] = {cam.C[0], cam.C[1], cam.C[2]};
  float dir[3];
  camera_ray(cam, h, w, dir);
  int face_idx = 0;
  float t, tu, tv;
  bool valid = ray_mesh(tile, orig, dir, in_img, verts, faces, n_faces, face_idx, t, tu, tv);
  valid = valid && in_img;
  if (in_img) {
    if (depth) depth[idx] = valid ? t : -1;
    color[idx * 3 + 0] = 0;
    color[idx * 3 + 1] = 0;
    color[idx * 3 + 2] = 0;
  }
  float pt[3] = {0.f, 0.f, 0.f}, pdir[3] = {0.f, 0.f, 1.f};
  const float porig[3] = {proj.C[0], proj.C[1], proj.C[2]};
  if (valid) {
    if (normal) {                                                // render.h:283-312
      const int* face = faces + (long)face_idx * 3;
      const float tw = 1 - tu - tv;
      const float *a = verts + (long)face[0] * 3, *b = verts + (long)face[1] * 3, *c = verts + (long)face[2] * 3;
      const float e1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, e2[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
      float nrm[3];
      cross3(e1, e2, nrm);
      normalize3(nrm, nrm);
      if (dot3(nrm, dir) > 0) { nrm[0] = nrm[0] * -1.f; nrm[1] = nrm[1] * -1.f; nrm[2] = nrm[2] * -1.f; }
      float col[3];
      bary_mix(colors, face, tu, tv, tw, col);
      const float refl = phong_at_camera(orig, dir, t, nrm, ka, kd, ks, alpha);
#pragma unroll
      for (int k = 0; k < 3; ++k) normal[idx * 3 + k] = std_min(1.f, std_max(0.f, refl * col[k]));
    }
    pt[0] = dir[0] * t; pt[1] = dir[1] * t; pt[2] = dir[2] * t;
    pt[0] = orig[0] + pt[0]; pt[1] = orig[1] + pt[1]; pt[2] = orig[2] + pt[2];
    pdir[0] = pt[0] - porig[0]; pdir[1] = pt[1] - porig[1]; pdir[2] = pt[2] - porig[2];
    const float z = pdir[2];
    pdir[0] = pdir[0] / z; pdir[1] = pdir[1] / z; pdir[2] = pdir[2] / z;
  }
  // shadow ray: does the projector see the same surface point?
  int p_face = 0;
  float p_t, p_tu, p_tv;
  const bool p_valid = ray_mesh(tile, porig, pdir, valid, verts, faces, n_faces, p_face, p_t, p_tu, p_tv);
  if (!valid || !p_valid) return;
  float p_pt[3] = {pdir[0] * p_t, pdir[1] * p_t, pdir[2] * p_t};
  p_pt[0] = porig[0] + p_pt[0]; p_pt[1] = porig[1] + p_pt[1]; p_pt[2] = porig[2] + p_pt[2];
  const float diff[3] = {p_pt[0] - pt[0], p_pt[1] - pt[1], p_pt[2] - pt[2]};
  if ((double)norm3(diff) > 1e-5) return;                        // float against a double literal, render.h:338
  float y[3];                                                    // Camera::to_2d, render.h:62-71
  y[0] = proj.R[0] * p_pt[0] + proj.R[1] * p_pt[1] + proj.R[2] * p_pt[2] + proj.t[0];
  y[1] = proj.R[3] * p_pt[0] + proj.R[4] * p_pt[1] + proj.R[5] * p_pt[2] + proj.t[1];
  y[2] = proj.R[6] * p_pt[0] + proj.R[7] * p_pt[1] + proj.R[8] * p_pt[2] + proj.t[2];
  float u = proj.fx * y[0] + proj.px * y[2];
  float v = proj.fy * y[1] + proj.py * y[2];
  const float d = y[2];
  u /= d;
  v /= d;
  if (u >= 0 && v >= 0 && u < proj.width && v < proj.height) {
    int x1 = (int)u, y1 = (int)v;                                // interpolate_linear, render.h:228-249
    int x2 = x1 + 1, y2 = y1 + 1;
    const float denom = (float)((x2 - x1) * (y2 - y1));
    const float t11 = (x2 - u) * (y2 - v);
    const float t21 = (u - x1) * (y2 - v);
    const float t12 = (x2 - u) * (v - y1);
    const float t22 = (u - x1) * (v - y1);
    x1 = min(max(x1, 0), proj.width - 1);
    x2 = min(max(x2, 0), proj.width - 1);
    y1 = min(max(y1, 0), proj.height - 1);
    y2 = min(max(y2, 0), proj.height - 1);
    float decay = d_alpha + d_beta * d;
    decay *= decay;
    decay = std_max(decay, 1.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float c = (pattern[((long)y1 * proj.width + x1) * 3 + k] * t11 + pattern[((long)y2 * proj.width + x1) * 3 + k] * t12 +
                       pattern[((long)y1 * proj.width + x2) * 3 + k] * t21 + pattern[((long)y2 * proj.width + x2) * 3 + k] * t22) /
                      denom;
      color[idx * 3 + k] = c / decay;
    }
  }
}

int render_mesh_proj_f32(const float* verts, const float* colors, const int* faces, int n_faces, const float* cam_p,
                         int cam_w, int cam_h, const float* proj_p, int proj_w, int proj_h, const float* shader,
                         const float* pattern, float d_alpha, float d_beta, float* depth, float* color, float* normal,
                         hipStream_t stream) {
  const CamDev cam = make_cam(cam_p, cam_w, cam_h), proj = make_cam(proj_p, proj_w, proj_h);
  const long n = (long)cam_w * cam_h;
  hipLaunchKernelGGL(render_proj_kernel, dim3((unsigned)ceil_div(n, 256L)), dim3(256), 0, stream, verts, colors, faces,
                     n_faces, cam, proj, shader[0], shader[1], shader[2], shader[3], pattern, d_alpha, d_beta, depth, color,
                     normal);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// RenderMeshFunctor<float>::operator() (render.h:150-223): camera rays only; the normal buffer receives the
// interpolated vertex normals (flipped towards the camera, not normalised), the colour buffer the Phong-shaded
// interpolated vertex colours.  Buffers may be null like the reference's.
__global__ __launch_bounds__(256) void render_mesh_kernel(const float* __restrict__ verts, const float* __restrict__ colors,
                                                          const float* __restrict__ normals, const int* __restrict__ faces,
                                                          int n_faces, CamDev cam, float ka, float kd, float ks, float alpha,
                                                          float* __restrict__ depth, float* __restrict__ color,
                                                          float* __restrict__ normal) {
  __shared__ float tile[kFaceTile * 3][4];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_img = idx < cam.width * cam.height;
  const int h = idx / cam.width, w = idx % cam.width;
  const float orig[3] = {cam.C[0], cam.C[1], cam.C[2]};
  float dir[3];
  camera_ray(cam, h, w, dir);
  int face_idx = 0;
  float t, tu, tv;
  const bool valid = ray_mesh(tile, orig, dir, in_img, verts, faces, n_faces, face_idx, t, tu, tv);
  if (!in_img) return;
  if (depth) depth[idx] = valid ? t : -1;
  if (!valid) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (color) color[idx * 3 + k] = 0;
      if (normal) normal[idx * 3 + k] = 0;
    }
    return;
  }
  if (!normal && !color) return;
  const int* face = faces + (long)face_idx * 3;
  const float tw = 1 - tu - tv;
  float nrm[3];
  bary_mix(normals, face, tu, tv, tw, nrm);
  if (dot3(nrm, dir) > 0) { nrm[0] = nrm[0] * -1.f; nrm[1] = nrm[1] * -1.f; nrm[2] = nrm[2] * -1.f; }
  if (normal) {
#pragma unroll
    for (int k = 0; k < 3; ++k) normal[idx * 3 + k] = nrm[k];
  }
  if (color) {
    float col[3];
    bary_mix(colors, face, tu, tv, tw, col);
    const float refl = phong_at_camera(orig, dir, t, nrm, ka, kd, ks, alpha);
#pragma unroll
    for (int k = 0; k < 3; ++k) color[idx * 3 + k] = std_min(1.f, std_max(0.f, refl * col[k]));
  }
}

int render_mesh_f32(const float* verts, const float* colors, const float* normals, const int* faces, int n_faces,
                    const float* cam_p, int cam_w, int cam_h, const float* shader, float* depth, float* color, float* normal,
                    hipStream_t stream) {
  const CamDev cam = make_cam(cam_p, cam_w, cam_h);
  const long n = (long)cam_w * cam_h;
  hipLaunchKernelGGL(render_mesh_kernel, dim3((unsigned)ceil_div(n, 256L)), dim3(256), 0, stream, verts, colors, normals,
                     faces, n_faces, cam, shader[0], shader[1], shader[2], shader[3], depth, color, normal);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
