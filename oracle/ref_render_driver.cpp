// Thin C driver around the reference's own CPU renderer (TEST INFRASTRUCTURE, build container only).
// oracle/build_ref.py compiles THIS file with -I /root/reference/renderer/render; the reference sources are
// included from where they lie (render_cpu.cpp instantiates RendererCpu<float>), nothing of them is copied.
#include "render_cpu.cpp"

extern "C" int ctd_ref_render_mesh_proj(const float* verts, const float* colors, const float* normals, int n_verts,
                                        const int* faces, int n_faces, const float* cam /* fx fy px py R[9] t[3] */,
                                        int cam_w, int cam_h, const float* proj, int proj_w, int proj_h,
                                        const float* shader /* ka kd ks alpha */, const float* pattern, float d_alpha,
                                        float d_beta, float* depth, float* color, float* normal, int n_threads) {
  RenderInput<float> in;
  in.verts = const_cast<float*>(verts);
  in.colors = const_cast<float*>(colors);
  in.normals = const_cast<float*>(normals);
  in.n_verts = n_verts;
  in.faces = const_cast<int*>(faces);
  in.n_faces = n_faces;
  Buffer<float> buf;
  buf.depth = depth;
  buf.color = color;
  buf.normal = normal;
  Camera<float> c(cam[0], cam[1], cam[2], cam[3], cam + 4, cam + 13, cam_w, cam_h);
  Camera<float> p(proj[0], proj[1], proj[2], proj[3], proj + 4, proj + 13, proj_w, proj_h);
  Shader<float> sh(shader[0], shader[1], shader[2], shader[3]);
  RendererCpu<float> r(c, sh, buf, n_threads);
  r.render_mesh_proj(in, p, pattern, d_alpha, d_beta);
  return 0;
}

extern "C" int ctd_ref_render_mesh(const float* verts, const float* colors, const float* normals, int n_verts,
                                   const int* faces, int n_faces, const float* cam, int cam_w, int cam_h,
                                   const float* shader, float* depth, float* color, float* normal, int n_threads) {
  RenderInput<float> in;
  in.verts = const_cast<float*>(verts);
  in.colors = const_cast<float*>(colors);
  in.normals = const_cast<float*>(normals);
  in.n_verts = n_verts;
  in.faces = const_cast<int*>(faces);
  in.n_faces = n_faces;
  Buffer<float> buf;
  buf.depth = depth;
  buf.color = color;
  buf.normal = normal;
  Camera<float> c(cam[0], cam[1], cam[2], cam[3], cam + 4, cam + 13, cam_w, cam_h);
  Shader<float> sh(shader[0], shader[1], shader[2], shader[3]);
  RendererCpu<float> r(c, sh, buf, n_threads);
  r.render_mesh(in);
  return 0;
}
