"""Mirror of the reference's `renderer` package (renderer/cyrender.pyx:80-200) on top of libctd_hip.so.

Same class names and call signatures as the Cython module the data generator uses
(data/create_syn_data.py:152-160):

    cam = PyCamera(fx, fy, px, py, R, t, width, height)
    data = PyRenderInput(verts=..., colors=..., normals=..., faces=...)
    r = PyRenderer(cam, PyShader(0.5, 1.5, 0.0, 10), engine='gpu')
    r.mesh_proj(data, proj, pattern, d_alpha=0, d_beta=0.35)
    im, depth, ambient = r.color(), r.depth(), r.normal()

numpy in, numpy out like the reference; `render_mesh_proj` / `render_mesh` below are the tensor-level calls that keep
everything on the device.  Both renderers of the reference are provided (`mesh_proj`, the one the data pipeline
uses, and the plain `mesh`); there is no CPU engine in this package (engine='cpu' raises: the CPU path lives on as
the test oracle).
"""
import numpy as np
import torch

from . import _lib


def _cam_params(fx, fy, px, py, R, t):
    R = np.asarray(R, np.float32)
    t = np.asarray(t, np.float32)
    if R.shape != (3, 3):
        raise Exception('invalid R matrix')                      # cyrender.pyx:85
    if t.shape != (3,):
        raise Exception('invalid t vector')
    return np.ascontiguousarray(np.concatenate([[fx, fy, px, py], R.reshape(9), t]).astype(np.float32))


class PyCamera:
    def __init__(self, fx, fy, px, py, R, t, width, height):
        self.params = _cam_params(fx, fy, px, py, R, t)
        self.width, self.height = int(width), int(height)


class PyShader:
    def __init__(self, ka, kd, ks, alpha):
        self.params = np.array([ka, kd, ks, alpha], np.float32)


class PyRenderInput:
    def __init__(self, verts=None, colors=None, normals=None, faces=None):
        self.verts = self.colors = self.normals = self.faces = None
        if verts is not None:
            self.set_verts(verts)
        if normals is not None:
            self.set_normals(normals)
        if colors is not None:
            self.set_colors(colors)
        if faces is not None:
            self.set_faces(faces)

    @staticmethod
    def _nx3(a, dtype, what):
        a = np.ascontiguousarray(a, dtype)
        if a.ndim != 2 or a.shape[1] != 3:
            raise Exception('%s has to be a Nx3 matrix' % what)
        return a

    def set_verts(self, verts):
        self.verts = self._nx3(verts, np.float32, 'verts')

    def set_colors(self, colors):
        self.colors = self._nx3(colors, np.float32, 'colors')

    def set_normals(self, normals):
        self.normals = self._nx3(normals, np.float32, 'normals')

    def set_faces(self, faces):
        self.faces = self._nx3(faces, np.int32, 'faces')


def render_mesh_proj(verts, colors, faces, cam, proj, shader, pattern, d_alpha=1.0, d_beta=0.0):
    """verts, colors [n,3] f32, faces [m,3] int32, pattern [ph,pw,3] f32: CUDA tensors; cam, proj: PyCamera; shader:
    PyShader -> (depth [H,W], color [H,W,3], normal [H,W,3]) CUDA tensors (normal zero where nothing is hit)."""
    for t_, name, dt in ((verts, "verts", torch.float32), (colors, "colors", torch.float32), (faces, "faces", torch.int32),
                         (pattern, "pattern", torch.float32)):
        if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.is_contiguous() and t_.dtype == dt):
            raise RuntimeError("%s must be a contiguous CUDA tensor of dtype %s" % (name, dt))
    if tuple(pattern.shape) != (proj.height, proj.width, 3):
        raise Exception('pattern has to be a %dx%dx3 tensor' % (proj.height, proj.width))      # cyrender.pyx:197
    dev = verts.device
    depth = torch.empty((cam.height, cam.width), dtype=torch.float32, device=dev)
    color = torch.empty((cam.height, cam.width, 3), dtype=torch.float32, device=dev)
    normal = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device=dev)
    st = _lib.lib().ctd_render_mesh_proj_f32(
        verts.data_ptr(), colors.data_ptr(), verts.shape[0], faces.data_ptr(), faces.shape[0],
        cam.params.ctypes.data, cam.width, cam.height, proj.params.ctypes.data, proj.width, proj.height,
        shader.params.ctypes.data, pattern.data_ptr(), float(d_alpha), float(d_beta), depth.data_ptr(), color.data_ptr(),
        normal.data_ptr(), dev.index, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "render_mesh_proj")
    return depth, color, normal


def render_mesh(verts, colors, normals, faces, cam, shader):
    """RenderMeshFunctor (render.h:150-223): verts, colors, normals [n,3] f32, faces [m,3] int32 CUDA tensors ->
    (depth [H,W], color [H,W,3], normal [H,W,3]) CUDA tensors."""
    for t_, name, dt in ((verts, "verts", torch.float32), (colors, "colors", torch.float32),
                         (normals, "normals", torch.float32), (faces, "faces", torch.int32)):
        if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.is_contiguous() and t_.dtype == dt):
            raise RuntimeError("%s must be a contiguous CUDA tensor of dtype %s" % (name, dt))
    if colors.shape != verts.shape or normals.shape != verts.shape:
        raise RuntimeError("colors and normals must have one row per vertex")
    dev = verts.device
    depth = torch.empty((cam.height, cam.width), dtype=torch.float32, device=dev)
    color = torch.empty((cam.height, cam.width, 3), dtype=torch.float32, device=dev)
    normal = torch.empty((cam.height, cam.width, 3), dtype=torch.float32, device=dev)
    st = _lib.lib().ctd_render_mesh_f32(
        verts.data_ptr(), colors.data_ptr(), normals.data_ptr(), verts.shape[0], faces.data_ptr(), faces.shape[0],
        cam.params.ctypes.data, cam.width, cam.height, shader.params.ctypes.data, depth.data_ptr(), color.data_ptr(),
        normal.data_ptr(), dev.index, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "render_mesh")
    return depth, color, normal


class PyRenderer:
    def __init__(self, cam, shader, engine='gpu', n_threads=1, device=None):
        if engine != 'gpu':
            raise Exception('invalid engine' if engine != 'cpu' else
                            "engine='cpu' is not part of this package (the CPU renderer is the test oracle)")
        self.cam, self.shader = cam, shader
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.depth_buffer = np.zeros((cam.height, cam.width), np.float32)
        self.color_buffer = np.zeros((cam.height, cam.width, 3), np.float32)
        self.normal_buffer = np.zeros((cam.height, cam.width, 3), np.float32)

    def depth(self):
        return self.depth_buffer

    def color(self):
        return self.color_buffer

    def normal(self):
        return self.normal_buffer

    def _store(self, d, c, n):
        self.depth_buffer[...] = d.cpu().numpy()
        self.color_buffer[...] = c.cpu().numpy()
        self.normal_buffer[...] = n.cpu().numpy()

    def mesh(self, input):
        up = lambda a: torch.from_numpy(a).to(self.device)
        self._store(*render_mesh(up(input.verts), up(input.colors), up(input.normals), up(input.faces), self.cam,
                                 self.shader))

    def mesh_proj(self, input, proj, pattern, d_alpha=1, d_beta=0):
        pattern = np.ascontiguousarray(pattern, np.float32)
        if pattern.shape != (proj.height, proj.width, 3):
            raise Exception('pattern has to be a %dx%dx3 tensor' % (proj.height, proj.width))
        up = lambda a: torch.from_numpy(a).to(self.device)
        d, c, n = render_mesh_proj(up(input.verts), up(input.colors), up(input.faces), self.cam, proj, self.shader,
                                   up(pattern), d_alpha, d_beta)
        self._store(d, c, n)
