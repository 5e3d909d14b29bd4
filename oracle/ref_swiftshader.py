"""TEST INFRASTRUCTURE -- container-only executable reference.

Runs the reference's OWN per-pixel code -- the GLSL-ES shaders under
/root/reference/src/figdraw/opengl/glsl/emscripten/ -- on the headless
SwiftShader GLES3 that ships inside the `kaleido` Python package, driven
through EGL pbuffers with ctypes.  Nothing from the reference is copied: the
shader text is read from /root/reference at run time, so this module is inert
on the GPU box (``available()`` returns False there).

It is used for exactly two things:
  * pinning ``oracle/figdraw_oracle.c`` (the C restatement) against real
    reference output, and
  * generating the golden vectors committed under ``tests/golden/``
    (``tools/make_goldens.py``).

The host side below restates, at the BackendContext call level, what
``src/figdraw/opengl/glcontext.nim`` does between ``beginFrame`` and
``readPixels``: quad emission (glcontext.nim:1449-1559), radii packing
(:745-817), blend state (utils/glutils.nim:150-154), masks (:1873-1949),
backdrop blur (:1743-1841), atlas packing (:541-586), readback (:2094-2135).
It is deliberately written in numpy float32 and shares no code with the C
oracle or with the HIP product, so agreement between the three is evidence.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

REF_ROOT = "/root/reference"
GLSL_DIR = os.path.join(REF_ROOT, "src/figdraw/opengl/glsl/emscripten")
_SS_DIR = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader"

f32 = np.float32


def available() -> bool:
    return os.path.isdir(GLSL_DIR) and os.path.exists(os.path.join(_SS_DIR, "libEGL.so"))


# --------------------------------------------------------------------------- GL/EGL plumbing
_egl = None
_gl = None

GL_FLOAT = 0x1406
GL_UNSIGNED_BYTE = 0x1401
GL_UNSIGNED_SHORT = 0x1403
GL_ARRAY_BUFFER = 0x8892
GL_ELEMENT_ARRAY_BUFFER = 0x8893
GL_STREAM_DRAW = 0x88E0
GL_STATIC_DRAW = 0x88E4
GL_TRIANGLES = 0x0004
GL_BLEND = 0x0BE2
GL_SRC_ALPHA = 0x0302
GL_ONE_MINUS_SRC_ALPHA = 0x0303
GL_ONE = 1
GL_COLOR_BUFFER_BIT = 0x4000
GL_TEXTURE_2D = 0x0DE1
GL_TEXTURE0 = 0x84C0
GL_RGBA = 0x1908
GL_RGBA8 = 0x8058
GL_TEXTURE_MIN_FILTER = 0x2801
GL_TEXTURE_MAG_FILTER = 0x2800
GL_TEXTURE_WRAP_S = 0x2802
GL_TEXTURE_WRAP_T = 0x2803
GL_LINEAR = 0x2601
GL_LINEAR_MIPMAP_LINEAR = 0x2703
GL_CLAMP_TO_EDGE = 0x812F
GL_FRAMEBUFFER = 0x8D40
GL_COLOR_ATTACHMENT0 = 0x8CE0
GL_FRAGMENT_SHADER = 0x8B30
GL_VERTEX_SHADER = 0x8B31
GL_COMPILE_STATUS = 0x8B81
GL_LINK_STATUS = 0x8B82
GL_PACK_ALIGNMENT = 0x0D05
GL_UNPACK_ALIGNMENT = 0x0CF5


def _load():
    global _egl, _gl
    if _gl is not None:
        return
    _egl = C.CDLL(os.path.join(_SS_DIR, "libEGL.so"), mode=C.RTLD_GLOBAL)
    _gl = C.CDLL(os.path.join(_SS_DIR, "libGLESv2.so"), mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    _egl.eglGetDisplay.restype = vp
    _egl.eglGetDisplay.argtypes = [vp]
    _egl.eglInitialize.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _egl.eglChooseConfig.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    _egl.eglCreatePbufferSurface.restype = vp
    _egl.eglCreatePbufferSurface.argtypes = [vp, vp, C.POINTER(C.c_int)]
    _egl.eglBindAPI.argtypes = [C.c_uint]
    _egl.eglCreateContext.restype = vp
    _egl.eglCreateContext.argtypes = [vp, vp, vp, C.POINTER(C.c_int)]
    _egl.eglMakeCurrent.argtypes = [vp, vp, vp, vp]
    _egl.eglDestroyContext.argtypes = [vp, vp]
    _egl.eglDestroySurface.argtypes = [vp, vp]
    g = _gl
    g.glCreateShader.restype = C.c_uint
    g.glCreateShader.argtypes = [C.c_uint]
    g.glShaderSource.argtypes = [C.c_uint, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    g.glCompileShader.argtypes = [C.c_uint]
    g.glGetShaderiv.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_int)]
    g.glGetShaderInfoLog.argtypes = [C.c_uint, C.c_int, C.POINTER(C.c_int), C.c_char_p]
    g.glCreateProgram.restype = C.c_uint
    g.glAttachShader.argtypes = [C.c_uint, C.c_uint]
    g.glLinkProgram.argtypes = [C.c_uint]
    g.glGetProgramiv.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_int)]
    g.glGetProgramInfoLog.argtypes = [C.c_uint, C.c_int, C.POINTER(C.c_int), C.c_char_p]
    g.glUseProgram.argtypes = [C.c_uint]
    g.glGetUniformLocation.restype = C.c_int
    g.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
    g.glGetAttribLocation.restype = C.c_int
    g.glGetAttribLocation.argtypes = [C.c_uint, C.c_char_p]
    g.glUniform1i.argtypes = [C.c_int, C.c_int]
    g.glUniform1f.argtypes = [C.c_int, C.c_float]
    g.glUniform2f.argtypes = [C.c_int, C.c_float, C.c_float]
    g.glUniformMatrix4fv.argtypes = [C.c_int, C.c_int, C.c_ubyte, C.POINTER(C.c_float)]
    g.glGenBuffers.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    g.glBindBuffer.argtypes = [C.c_uint, C.c_uint]
    g.glBufferData.argtypes = [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint]
    g.glEnableVertexAttribArray.argtypes = [C.c_uint]
    g.glDisableVertexAttribArray.argtypes = [C.c_uint]
    g.glVertexAttribPointer.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p]
    g.glDrawElements.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_void_p]
    g.glDrawArrays.argtypes = [C.c_uint, C.c_int, C.c_int]
    g.glViewport.argtypes = [C.c_int] * 4
    g.glClearColor.argtypes = [C.c_float] * 4
    g.glClear.argtypes = [C.c_uint]
    g.glEnable.argtypes = [C.c_uint]
    g.glDisable.argtypes = [C.c_uint]
    g.glBlendFuncSeparate.argtypes = [C.c_uint] * 4
    g.glGenTextures.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    g.glBindTexture.argtypes = [C.c_uint, C.c_uint]
    g.glActiveTexture.argtypes = [C.c_uint]
    g.glTexImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
    g.glTexSubImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
    g.glTexParameteri.argtypes = [C.c_uint, C.c_uint, C.c_int]
    g.glGenerateMipmap.argtypes = [C.c_uint]
    g.glCopyTexSubImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    g.glGenFramebuffers.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    g.glBindFramebuffer.argtypes = [C.c_uint, C.c_uint]
    g.glFramebufferTexture2D.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int]
    g.glReadPixels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
    g.glPixelStorei.argtypes = [C.c_uint, C.c_int]
    g.glFinish.argtypes = []
    g.glGetError.restype = C.c_uint


def _read(name: str) -> bytes:
    with open(os.path.join(GLSL_DIR, name), "rb") as f:
        return f.read()


def _compile(kind: int, src: bytes) -> int:
    sh = _gl.glCreateShader(kind)
    p = C.c_char_p(src)
    _gl.glShaderSource(sh, 1, C.byref(p), None)
    _gl.glCompileShader(sh)
    ok = C.c_int(0)
    _gl.glGetShaderiv(sh, GL_COMPILE_STATUS, C.byref(ok))
    if not ok.value:
        buf = C.create_string_buffer(8192)
        _gl.glGetShaderInfoLog(sh, 8192, None, buf)
        raise RuntimeError("shader compile failed: " + buf.value.decode())
    return sh


class _Program:
    def __init__(self, vert: str, frag: str):
        self.id = _gl.glCreateProgram()
        _gl.glAttachShader(self.id, _compile(GL_VERTEX_SHADER, _read(vert)))
        _gl.glAttachShader(self.id, _compile(GL_FRAGMENT_SHADER, _read(frag)))
        _gl.glLinkProgram(self.id)
        ok = C.c_int(0)
        _gl.glGetProgramiv(self.id, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            buf = C.create_string_buffer(8192)
            _gl.glGetProgramInfoLog(self.id, 8192, None, buf)
            raise RuntimeError("program link failed: " + buf.value.decode())

    def u(self, name: str) -> int:
        return _gl.glGetUniformLocation(self.id, name.encode())

    def a(self, name: str) -> int:
        return _gl.glGetAttribLocation(self.id, name.encode())


# --------------------------------------------------------------------------- small f32 matrix helpers
def _mat_identity():
    return np.eye(4, dtype=f32)


def _mat_translate(x, y):
    m = np.eye(4, dtype=f32)
    m[0, 3] = f32(x)
    m[1, 3] = f32(y)
    return m


def _mat_scale(x, y):
    m = np.eye(4, dtype=f32)
    m[0, 0] = f32(x)
    m[1, 1] = f32(y)
    return m


def _mat_rotate_z(angle):
    # vmath rotateZ(angle) (glcontext.nim:1995-1997): m[0,1] = -sin, m[1,0] = sin in vmath's [column, row]
    # indexing, i.e. x' = cos x + sin y, y' = -sin x + cos y.  Pinned by tests/expected/render_line_rect.png.
    c = f32(math.cos(float(f32(angle))))
    s = f32(math.sin(float(f32(angle))))
    m = np.eye(4, dtype=f32)
    m[0, 0] = c
    m[0, 1] = s
    m[1, 0] = -s
    m[1, 1] = c
    return m


def _round_half_away(x):
    # Nim's math.round: half away from zero
    x = f32(x)
    return f32(math.floor(float(x) + 0.5)) if x >= 0 else f32(-math.floor(-float(x) + 0.5))


def _clamp_radius(r, m):
    r = f32(r)
    if r <= 0:
        return f32(0)
    return _round_half_away(max(f32(1), min(r, f32(m))))


def rounded_radii_vec(rx, ry, hx, hy):
    """glcontext.nim:751-817.  rx/ry in order TL,TR,BL,BR; returns (r4, elliptical)."""
    TL, TR, BL, BR = 0, 1, 2, 3
    hx = f32(hx)
    hy = f32(hy)
    if all(f32(rx[i]) == f32(ry[i]) for i in range(4)):
        m = min(hx, hy)
        c = [_clamp_radius(rx[i], m) for i in range(4)]
        return np.array([c[TR], c[BR], c[TL], c[BL]], dtype=f32), False
    cx = [_clamp_radius(rx[i], hx) for i in range(4)]
    cy = [_clamp_radius(ry[i], hy) for i in range(4)]
    cm = min(hx, hy)

    def pack(x, y):
        qx = _round_half_away(f32(min(max(f32(x) / max(hx, f32(0.000001)), f32(0)), f32(1))) * f32(4095))
        qy = _round_half_away(f32(min(max(f32(y) / max(hy, f32(0.000001)), f32(0)), f32(1))) * f32(4095))
        return f32(qx + qy * f32(4096))

    def enc(i):
        if f32(rx[i]) == f32(ry[i]):
            return f32(-(_clamp_radius(rx[i], cm) + f32(1)))
        if cx[i] == cy[i]:
            return f32(-(cx[i] + f32(1)))
        return pack(cx[i], cy[i])

    return np.array([enc(TR), enc(BR), enc(TL), enc(BL)], dtype=f32), True


# --------------------------------------------------------------------------- the reference-side backend
_ATTRS = [
    # name, components, gl type, normalized
    ("vertexPos", 2, GL_FLOAT, 0),
    ("vertexUv", 2, GL_FLOAT, 0),
    ("vertexColor", 4, GL_UNSIGNED_BYTE, 1),
    ("vertexFillMidColor", 4, GL_UNSIGNED_BYTE, 1),
    ("vertexFillStopColor", 4, GL_UNSIGNED_BYTE, 1),
    ("vertexSdfParams", 4, GL_FLOAT, 0),
    ("vertexSdfRadii", 4, GL_FLOAT, 0),
    ("vertexSdfMode", 1, GL_FLOAT, 0),
    ("vertexSdfFactors", 2, GL_FLOAT, 0),
    ("vertexSubpixelShift", 1, GL_FLOAT, 0),
    ("vertexRectMaskParams", 4, GL_FLOAT, 0),
    ("vertexRectMaskRadii", 4, GL_FLOAT, 0),
    ("vertexRectMaskMatX", 4, GL_FLOAT, 0),
    ("vertexRectMaskMatY", 4, GL_FLOAT, 0),
]


class RefGL:
    """A BackendContext-shaped driver for the reference shaders on SwiftShader."""

    def __init__(self, width: int, height: int, atlas_size: int = 1024, pixel_scale: float = 1.0):
        if not available():
            raise RuntimeError("reference shaders / SwiftShader not available here")
        _load()
        self.W, self.H = int(width), int(height)
        self.atlas_size = int(atlas_size)
        self.atlas_margin = 4
        self.pixel_scale = f32(pixel_scale)
        vp = C.c_void_p
        self.dpy = _egl.eglGetDisplay(None)
        maj, mnr = C.c_int(), C.c_int()
        assert _egl.eglInitialize(self.dpy, C.byref(maj), C.byref(mnr))
        attrs = (C.c_int * 13)(0x3033, 0x0001, 0x3040, 0x0004, 0x3024, 8, 0x3023, 8, 0x3022, 8, 0x3021, 8, 0x3038)
        cfg = vp()
        n = C.c_int()
        assert _egl.eglChooseConfig(self.dpy, attrs, C.byref(cfg), 1, C.byref(n)) and n.value >= 1
        sattr = (C.c_int * 5)(0x3057, self.W, 0x3056, self.H, 0x3038)
        self.surf = _egl.eglCreatePbufferSurface(self.dpy, cfg, sattr)
        assert self.surf
        _egl.eglBindAPI(0x30A0)
        cattr = (C.c_int * 3)(0x3098, 3, 0x3038)
        self.ctx = _egl.eglCreateContext(self.dpy, cfg, None, cattr)
        assert self.ctx
        assert _egl.eglMakeCurrent(self.dpy, self.surf, self.surf, self.ctx)

        g = _gl
        self.main = _Program("atlas.vert", "atlas.frag")
        self.rmask = _Program("atlas_rect_mask.vert", "atlas_rect_mask.frag")
        self.mask = _Program("atlas.vert", "mask.frag")
        self.blur = _Program("blur.vert", "blur.frag")
        self.vbos = {}
        for name, *_ in _ATTRS:
            b = C.c_uint()
            g.glGenBuffers(1, C.byref(b))
            self.vbos[name] = b.value
        ib = C.c_uint()
        g.glGenBuffers(1, C.byref(ib))
        self.ibo = ib.value
        idx = np.array([3, 0, 1, 2, 3, 1], dtype=np.uint16)  # glcontext.nim:418-429
        g.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, self.ibo)
        g.glBufferData(GL_ELEMENT_ARRAY_BUFFER, idx.nbytes, idx.ctypes.data, GL_STATIC_DRAW)
        bb = (C.c_uint * 2)()
        g.glGenBuffers(2, bb)
        self.blur_pos, self.blur_uv = bb[0], bb[1]
        # full-screen triangle (glcontext.nim blur VAO setup): pos (-1,-1),(3,-1),(-1,3); uv (0,0),(2,0),(0,2)
        p = np.array([-1, -1, 3, -1, -1, 3], dtype=f32)
        u = np.array([0, 0, 2, 0, 0, 2], dtype=f32)
        g.glBindBuffer(GL_ARRAY_BUFFER, self.blur_pos)
        g.glBufferData(GL_ARRAY_BUFFER, p.nbytes, p.ctypes.data, GL_STATIC_DRAW)
        g.glBindBuffer(GL_ARRAY_BUFFER, self.blur_uv)
        g.glBufferData(GL_ARRAY_BUFFER, u.nbytes, u.ctypes.data, GL_STATIC_DRAW)

        # startOpenGL blend state (utils/glutils.nim:150-154)
        g.glEnable(GL_BLEND)
        g.glBlendFuncSeparate(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA, GL_ONE, GL_ONE_MINUS_SRC_ALPHA)
        g.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        g.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)

        self.atlas_tex = self._new_tex(self.atlas_size, self.atlas_size, GL_LINEAR_MIPMAP_LINEAR, GL_LINEAR, clamp=False, mip=True)
        self.heights = np.zeros(self.atlas_size, dtype=np.int64)
        self.entries = {}  # key -> (x, y, w, h) in UV units (f32)
        self.backdrop_tex = self._new_tex(self.W, self.H, GL_LINEAR, GL_LINEAR, clamp=True)
        self.backdrop_tmp = self._new_tex(self.W, self.H, GL_LINEAR, GL_LINEAR, clamp=True)
        self.mask_texs = [None]  # index 0 = "white", never sampled (maskTexEnabled false)
        fb = (C.c_uint * 2)()
        g.glGenFramebuffers(2, fb)
        self.mask_fbo, self.blur_fbo = fb[0], fb[1]

        self.mat = _mat_identity()
        self.mats = []
        self.aa = f32(1.2)
        self.mask_write = 0
        self.mask_begun = False
        self.rect_masks = []  # entries: ("fast", params, radii, matX, matY) | ("mask",)
        self.subpixel_enabled = False
        self.subpixel_shift = f32(0)
        self.frame_begun = False

    # ---- resources
    def _new_tex(self, w, h, minf, magf, clamp, mip=False):
        g = _gl
        t = C.c_uint()
        g.glGenTextures(1, C.byref(t))
        g.glBindTexture(GL_TEXTURE_2D, t.value)
        zeros = np.zeros((h, w, 4), dtype=np.uint8)
        g.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, w, h, 0, GL_RGBA, GL_UNSIGNED_BYTE, zeros.ctypes.data)
        g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, magf)
        g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, minf)
        if clamp:
            g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE)
            g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE)
        if mip:
            g.glGenerateMipmap(GL_TEXTURE_2D)
        return t.value

    def close(self):
        _egl.eglMakeCurrent(self.dpy, None, None, None)
        _egl.eglDestroyContext(self.dpy, self.ctx)
        _egl.eglDestroySurface(self.dpy, self.surf)

    # ---- atlas (glcontext.nim:541-586, textures.nim:106-119)
    def _find_empty_rect(self, w, h):
        iw = w + self.atlas_margin * 2
        ih = h + self.atlas_margin * 2
        lowest = self.atlas_size
        at = 0
        for i in range(self.atlas_size):
            v = int(self.heights[i])
            if v < lowest:
                fit = True
                for j in range(iw + 1):
                    if i + j >= self.atlas_size or int(self.heights[i + j]) > v:
                        fit = False
                        break
                if fit:
                    lowest = v
                    at = i
        if lowest + ih > self.atlas_size:
            raise RuntimeError("reference atlas full (grow not modelled in the harness)")
        self.heights[at:at + iw] = lowest + ih + self.atlas_margin * 2
        return at + self.atlas_margin, lowest + self.atlas_margin, w, h

    @staticmethod
    def minify_by2(img: np.ndarray) -> np.ndarray:
        """pixie Image.minifyBy2 (third-party, pixie >= 5.0.1) on premultiplied RGBA8, with the arithmetic the reference's own
        data/img1.flippy pins (its stored levels are this chain; tests/test_oracle.py::test_minify_by2_reproduces_the_flippy_levels):
        2x2 box SUM div 4; an odd extent rounds the result size up, the extra column / row is mix(a, b, 0.5) * 0.5 of the last
        source column / row (mix = (127 a + 128 b) div 255, * 0.5 = (128 v) div 255), the extra corner the last texel * 0.25."""
        img = img.astype(np.uint32)
        h, w = img.shape[:2]
        eh, ew = h // 2, w // 2
        out = np.zeros(((h + 1) // 2, (w + 1) // 2, 4), np.uint32)
        out[:eh, :ew] = (img[0:2 * eh:2, 0:2 * ew:2] + img[0:2 * eh:2, 1:2 * ew:2] + img[1:2 * eh:2, 1:2 * ew:2] + img[1:2 * eh:2, 0:2 * ew:2]) // 4
        half = lambda p, q: ((p * 127 + q * 128) // 255) * 128 // 255
        if w % 2:
            out[:eh, -1] = half(img[0:2 * eh:2, w - 1], img[1:2 * eh:2, w - 1])
        if h % 2:
            out[-1, :ew] = half(img[h - 1, 0:2 * ew:2], img[h - 1, 1:2 * ew:2])
            if w % 2:
                out[-1, -1] = img[h - 1, w - 1] * 64 // 255
        return out.astype(np.uint8)

    def put_image(self, key, rgba: np.ndarray):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = rgba.shape[:2]
        x, y, _, _ = self._find_empty_rect(w, h)
        s = f32(self.atlas_size)
        self.entries[key] = (f32(x) / s, f32(y) / s, f32(w) / s, f32(h) / s)
        g = _gl
        g.glBindTexture(GL_TEXTURE_2D, self.atlas_tex)
        img, lx, ly, level = rgba, x, y, 0
        while img.shape[1] > 1 and img.shape[0] > 1:
            img = np.ascontiguousarray(img)
            g.glTexSubImage2D(GL_TEXTURE_2D, level, lx, ly, img.shape[1], img.shape[0], GL_RGBA, GL_UNSIGNED_BYTE, img.ctypes.data)
            img = self.minify_by2(img)
            lx //= 2
            ly //= 2
            level += 1
        return x, y, w, h

    # ---- transforms (glcontext.nim:1991-2017)
    def save_transform(self):
        self.mats.append(self.mat.copy())

    def restore_transform(self):
        self.mat = self.mats.pop()

    def translate(self, x, y):
        self.mat = (self.mat @ _mat_translate(x, y)).astype(f32)

    def rotate(self, angle):
        self.mat = (self.mat @ _mat_rotate_z(angle)).astype(f32)

    def scale(self, sx, sy=None):
        self.mat = (self.mat @ _mat_scale(sx, sx if sy is None else sy)).astype(f32)

    def apply_transform(self, m16):
        # m16: column-major 16 floats (vmath Mat4 memory order)
        m = np.array(m16, dtype=f32).reshape(4, 4).T
        self.mat = (self.mat @ m).astype(f32)

    def set_text_subpixel_shift(self, shift):  # setTextSubpixelShift figbackend.nim:663-686
        self.subpixel_shift = f32(shift)

    def set_aa_factor(self, aa):
        self.aa = f32(aa)

    def _xf(self, x, y):
        m = self.mat
        x, y = f32(x), f32(y)
        return (f32(m[0, 0] * x + m[0, 1] * y + m[0, 3]), f32(m[1, 0] * x + m[1, 1] * y + m[1, 3]))

    # ---- frame (glcontext.nim:2080-2092, 1951-1980)
    def begin_frame(self, clear=True, color=(1.0, 1.0, 1.0, 1.0)):
        g = _gl
        g.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        g.glViewport(0, 0, self.W, self.H)
        if clear:
            g.glClearColor(*[float(c) for c in color])
            g.glClear(GL_COLOR_BUFFER_BIT)
        self.frame_begun = True
        self.rect_masks = []
        # ortho(0, w, h, 0, -1000, 1000), column-major for GL
        w, h = f32(self.W), f32(self.H)
        p = np.zeros((4, 4), dtype=f32)
        p[0, 0] = f32(2) / w
        p[1, 1] = f32(2) / (f32(0) - h)
        p[2, 2] = f32(-2) / f32(2000)
        p[0, 3] = -(w + f32(0)) / (w - f32(0))
        p[1, 3] = -(f32(0) + h) / (f32(0) - h)
        p[2, 3] = f32(0)
        p[3, 3] = f32(1)
        self.proj = np.ascontiguousarray(p.T)  # column-major memory

    def end_frame(self):
        assert self.mask_write == 0 and not self.rect_masks
        self.frame_begun = False
        _gl.glFinish()

    # ---- quad emission
    def _draw_quad(self, pos4, uv4, colors4, mid, stop, params, radii, mode_word, factors, mask_read=None):
        g = _gl
        if self.mask_begun:
            prog = self.mask
        else:
            fast = next((r for r in reversed(self.rect_masks) if r[0] == "fast"), None)
            prog = self.rmask if fast is not None else self.main
        if mask_read is None:
            mask_read = self.mask_write - 1 if self.mask_begun else self.mask_write
        g.glUseProgram(prog.id)
        data = {
            "vertexPos": np.array(pos4, dtype=f32),
            "vertexUv": np.array(uv4, dtype=f32),
            "vertexColor": np.array(colors4, dtype=np.uint8).reshape(4, 4),
            "vertexFillMidColor": np.tile(np.array(mid, dtype=np.uint8), (4, 1)),
            "vertexFillStopColor": np.tile(np.array(stop, dtype=np.uint8), (4, 1)),
            "vertexSdfParams": np.tile(np.array(params, dtype=f32), (4, 1)),
            "vertexSdfRadii": np.tile(np.array(radii, dtype=f32), (4, 1)),
            # +0.25: SwiftShader interpolates a per-quad constant float varying inexactly on
            # non-axis-aligned triangles (12.0 -> 11.999999 -> int() = 11 = a different mode).
            # The bias keeps int(sdfMode) and floor(sdfMode / 256) unchanged; LLVMpipe's plane
            # equations (zero gradients) are exact and need no such help.  Shader text untouched.
            "vertexSdfMode": np.full((4, 1), f32(mode_word) + f32(0.25), dtype=f32),
            "vertexSdfFactors": np.tile(np.array(factors, dtype=f32), (4, 1)),
            "vertexSubpixelShift": np.full((4, 1), self._active_subpixel_shift(), dtype=f32),
        }
        if prog is self.rmask:
            _, rp, rr, mx, my = fast
            data["vertexRectMaskParams"] = np.tile(np.array(rp, dtype=f32), (4, 1))
            data["vertexRectMaskRadii"] = np.tile(np.array(rr, dtype=f32), (4, 1))
            data["vertexRectMaskMatX"] = np.tile(np.array(mx, dtype=f32), (4, 1))
            data["vertexRectMaskMatY"] = np.tile(np.array(my, dtype=f32), (4, 1))
        enabled = []
        for name, comps, typ, norm in _ATTRS:
            loc = prog.a(name)
            if loc < 0 or name not in data:
                continue
            arr = np.ascontiguousarray(data[name])
            g.glBindBuffer(GL_ARRAY_BUFFER, self.vbos[name])
            g.glBufferData(GL_ARRAY_BUFFER, arr.nbytes, arr.ctypes.data, GL_STREAM_DRAW)
            g.glEnableVertexAttribArray(loc)
            g.glVertexAttribPointer(loc, comps, typ, norm, 0, None)
            enabled.append(loc)
        if prog.u("windowFrame") >= 0:
            g.glUniform2f(prog.u("windowFrame"), float(self.W), float(self.H))
        g.glUniformMatrix4fv(prog.u("proj"), 1, 0, self.proj.ctypes.data_as(C.POINTER(C.c_float)))
        if prog.u("aaFactor") >= 0:
            g.glUniform1f(prog.u("aaFactor"), float(self.aa))
        if prog.u("maskTexEnabled") >= 0:
            g.glUniform1i(prog.u("maskTexEnabled"), 1 if mask_read != 0 else 0)
        if prog.u("atlasTexelSize") >= 0:
            t = 1.0 / max(float(self.atlas_size), 1.0)
            g.glUniform2f(prog.u("atlasTexelSize"), t, t)
        if prog.u("subpixelPositioningEnabled") >= 0:
            g.glUniform1i(prog.u("subpixelPositioningEnabled"), 1 if self.subpixel_enabled else 0)
        if prog.u("atlasTex") >= 0:
            g.glActiveTexture(GL_TEXTURE0)
            g.glBindTexture(GL_TEXTURE_2D, self.atlas_tex)
            g.glUniform1i(prog.u("atlasTex"), 0)
        if prog.u("maskTex") >= 0 and mask_read != 0:
            g.glActiveTexture(GL_TEXTURE0 + 1)
            g.glBindTexture(GL_TEXTURE_2D, self.mask_texs[mask_read])
            g.glUniform1i(prog.u("maskTex"), 1)
        if prog.u("backdropTex") >= 0:
            g.glActiveTexture(GL_TEXTURE0 + 2)
            g.glBindTexture(GL_TEXTURE_2D, self.backdrop_tex)
            g.glUniform1i(prog.u("backdropTex"), 2)
        g.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, self.ibo)
        g.glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_SHORT, None)
        for loc in enabled:
            g.glDisableVertexAttribArray(loc)

    def _active_subpixel_shift(self):
        if not self.subpixel_enabled:
            return f32(0)
        return max(f32(0), min(self.subpixel_shift, f32(0.999)))

    def _quad_pos(self, x0, y0, x1, y1):
        """ceil(ctx.mat * corner), vertex order BL, BR, TR, TL (glcontext.nim:1498-1503)."""
        c = [self._xf(x0, y1), self._xf(x1, y1), self._xf(x1, y0), self._xf(x0, y0)]
        return [(f32(math.ceil(float(a))), f32(math.ceil(float(b)))) for a, b in c]

    # ---- drawRoundedRectSdf (glcontext.nim:1449-1617)
    def draw_rounded_rect_sdf(self, rect, colors, radii_x, radii_y, mode, factor=4.0, spread=0.0,
                              shape=(0.0, 0.0), fill_mode=0, mid=(0, 0, 0, 0), stop=(0, 0, 0, 0), mid_pos=0.5):
        x, y, w, h = [f32(v) for v in rect]
        if w <= 0 or h <= 0:
            return
        qhx, qhy = f32(w * f32(0.5)), f32(h * f32(0.5))
        inset = mode == 9
        sx, sy = f32(shape[0]), f32(shape[1])
        if sx > 0 and sy > 0:
            rsx, rsy = sx, sy
        else:
            rsx, rsy = w, h
        if inset:
            shx, shy = qhx, qhy
            params = (qhx, qhy, sx, sy)
        else:
            shx, shy = f32(rsx * f32(0.5)), f32(rsy * f32(0.5))
            params = (qhx, qhy, shx, shy)
        r4, ellip = rounded_radii_vec(radii_x, radii_y, shx, shy)
        pos = self._quad_pos(x, y, f32(x + w), f32(y + h))
        uv = [(0, 1), (1, 1), (1, 0), (0, 0)]
        if fill_mode == 0:
            factors = (f32(factor), f32(spread))
        else:
            factors = (f32(factor), min(max(f32(mid_pos), f32(0.01)), f32(0.99)))
        word = int(mode) + (128 if ellip else 0) + 256 * int(fill_mode)
        self._draw_quad(pos, uv, colors, mid, stop, params, r4, word, factors)

    # ---- images / msdf (glcontext.nim:1022-1155, 1169-1367)
    def draw_image(self, key, pos, colors, size=(0.0, 0.0), flip_y=False):
        if key not in self.entries:
            return
        ex, ey, ew, eh = self.entries[key]
        s = f32(self.atlas_size)
        if f32(size[0]) > 0 and f32(size[1]) > 0:
            dw, dh = f32(size[0]), f32(size[1])
        else:
            dw, dh = f32(ew * s), f32(eh * s)
        if flip_y:
            uv_at, uv_to = (ex, f32(ey + eh)), (f32(ex + ew), ey)
        else:
            uv_at, uv_to = (ex, ey), (f32(ex + ew), f32(ey + eh))
        px, py = f32(pos[0]), f32(pos[1])
        p4 = self._quad_pos(px, py, f32(px + dw), f32(py + dh))
        uv = [(uv_at[0], uv_to[1]), (uv_to[0], uv_to[1]), (uv_to[0], uv_at[1]), (uv_at[0], uv_at[1])]
        z4 = (0, 0, 0, 0)
        self._draw_quad(p4, uv, colors, z4, z4, z4, z4, 0, (0, 0))

    def draw_msdf(self, key, pos, color, size, px_range, sd_threshold=0.5, stroke_weight=0.0, mtsdf=False, flip_y=False):
        if key not in self.entries:
            return
        ex, ey, ew, eh = self.entries[key]
        if flip_y:
            uv_at, uv_to = (ex, f32(ey + eh)), (f32(ex + ew), ey)
        else:
            uv_at, uv_to = (ex, ey), (f32(ex + ew), f32(ey + eh))
        sw = max(f32(0), f32(stroke_weight))
        params = (f32(self.atlas_size), sw, 0, 0)
        if mtsdf:
            mode = 16 if sw > 0 else 14
        else:
            mode = 15 if sw > 0 else 13
        px, py = f32(pos[0]), f32(pos[1])
        p4 = self._quad_pos(px, py, f32(px + f32(size[0])), f32(py + f32(size[1])))
        uv = [(uv_at[0], uv_to[1]), (uv_to[0], uv_to[1]), (uv_to[0], uv_at[1]), (uv_at[0], uv_at[1])]
        z4 = (0, 0, 0, 0)
        self._draw_quad(p4, uv, [color] * 4, z4, z4, params, z4, mode, (f32(px_range), f32(sd_threshold)))

    # ---- drawable path: drawQuadraticBezierSdf (glcontext.nim:1619-1741), drawFilledQuad (:963-982), drawRect (:1410-1426)
    RECT_IMAGE_KEY = 0x7265637452454354

    @staticmethod
    def _lerp_color(a, b, t):
        ct = min(max(f32(t), f32(0)), f32(1))
        it = f32(1) - ct
        return tuple(int(_round_half_away(f32(a[i]) * it + f32(b[i]) * ct)) for i in range(4))

    def _gradient_colors(self, fill):
        """figbackend.nim:129-183 (vertex order BL, BR, TR, TL)."""
        kind, axis = fill["kind"], fill["axis"]
        if kind == 0:
            return [tuple(fill["start"])] * 4, 0.5

        mid = min(max(f32(fill["mid_pos"]) / f32(255), f32(0.01)), f32(0.99))

        def sample(t):
            if kind == 1:
                return self._lerp_color(fill["start"], fill["stop"], t)
            ct = min(max(f32(t), f32(0)), f32(1))
            if ct <= mid:
                return self._lerp_color(fill["start"], fill["mid"], ct / mid)
            return self._lerp_color(fill["mid"], fill["stop"], (ct - mid) / (f32(1) - mid))

        T = {0: (0, 1, 1, 0), 1: (1, 1, 0, 0), 2: (0.5, 1, 0.5, 0), 3: (0, 0.5, 1, 0.5)}[axis]
        return [sample(t) for t in T], mid

    def draw_quadratic_bezier_sdf(self, rect, fill, p0, p1, p2, stroke_weight, cap):
        x, y, w, h = [f32(v) for v in rect]
        if w <= 0 or h <= 0 or f32(stroke_weight) <= 0:
            return
        params = (f32(w * f32(0.5)), f32(h * f32(0.5)), f32(p0[0]), f32(p0[1]))
        curve = (f32(p1[0]), f32(p1[1]), f32(p2[0]), f32(p2[1]))
        pos = self._quad_pos(x, y, f32(x + w), f32(y + h))
        uv = [(0, 1), (1, 1), (1, 0), (0, 0)]
        mode = {2: 19, 3: 20}.get(int(cap), 18)
        z4 = (0, 0, 0, 0)
        if fill["kind"] == 2:
            mid = min(max(f32(fill["mid_pos"]) / f32(255), f32(0.01)), f32(0.99))
            fm = 1 + int(fill["axis"])
            self._draw_quad(pos, uv, [tuple(fill["start"])] * 4, tuple(fill["mid"]), tuple(fill["stop"]), params, curve,
                            mode + 256 * fm, (f32(stroke_weight), mid))
        else:
            cols, _ = self._gradient_colors(fill)
            self._draw_quad(pos, uv, cols, z4, z4, params, curve, mode, (f32(stroke_weight), f32(0)))

    def _rect_entry(self):
        if self.RECT_IMAGE_KEY not in self.entries:
            self.put_image(self.RECT_IMAGE_KEY, np.full((4, 4, 4), 255, dtype=np.uint8))
        return self.entries[self.RECT_IMAGE_KEY]

    def draw_filled_quad(self, verts, colors):
        ex, ey, ew, eh = self._rect_entry()
        u, v = f32(ex + ew / f32(2)), f32(ey + eh / f32(2))
        pos = []
        for i in range(4):
            a, b = self._xf(verts[2 * i], verts[2 * i + 1])
            pos.append((f32(math.ceil(float(a))), f32(math.ceil(float(b)))))
        z4 = (0, 0, 0, 0)
        self._draw_quad(pos, [(u, v)] * 4, colors, z4, z4, z4, z4, 0, (0, 0))

    def draw_rect(self, rect, color):
        ex, ey, ew, eh = self._rect_entry()
        u, v = f32(ex + ew / f32(2)), f32(ey + eh / f32(2))
        x, y, w, h = [f32(t) for t in rect]
        pos = self._quad_pos(x, y, f32(x + w), f32(y + h))
        z4 = (0, 0, 0, 0)
        self._draw_quad(pos, [(u, v)] * 4, [color] * 4, z4, z4, z4, z4, 0, (0, 0))

    # ---- masks (glcontext.nim:1873-1949)
    def begin_mask(self, rect, radii_x, radii_y):
        assert self.frame_begun and not self.mask_begun
        g = _gl
        self.mask_begun = True
        self.mask_write += 1
        if self.mask_write >= len(self.mask_texs):
            self.mask_texs.append(self._new_tex(self.W, self.H, GL_LINEAR, GL_LINEAR, clamp=False))
        g.glBindFramebuffer(GL_FRAMEBUFFER, self.mask_fbo)
        g.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, self.mask_texs[self.mask_write], 0)
        g.glViewport(0, 0, self.W, self.H)
        g.glClearColor(0, 0, 0, 0)
        g.glClear(GL_COLOR_BUFFER_BIT)
        red = (255, 0, 0, 255)
        self.draw_rounded_rect_sdf(rect, [red] * 4, radii_x, radii_y, 3, 4.0, 0.0)

    def end_mask(self):
        assert self.mask_begun
        self.mask_begun = False
        _gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)

    def pop_mask(self):
        self.mask_write -= 1

    def begin_rect_mask(self, rect, radii_x, radii_y):
        assert not self.mask_begun
        x, y, w, h = [f32(v) for v in rect]
        if not self.rect_masks and w > 0 and h > 0:
            hx, hy = f32(w * f32(0.5)), f32(h * f32(0.5))
            cx, cy = f32(x + hx), f32(y + hy)
            inv = np.linalg.inv(self.mat.astype(np.float64)).astype(f32)
            r4, ellip = rounded_radii_vec(radii_x, radii_y, hx, hy)
            self.rect_masks.append((
                "fast", (cx, cy, hx, hy), r4,
                (inv[0, 0], inv[0, 1], inv[0, 3], 1.0),
                (inv[1, 0], inv[1, 1], inv[1, 3], 1.0 if ellip else 0.0),
            ))
        else:
            self.begin_mask(rect, radii_x, radii_y)
            self.end_mask()
            self.rect_masks.append(("mask",))

    def pop_rect_mask(self):
        rm = self.rect_masks.pop()
        if rm[0] == "mask":
            self.pop_mask()

    # ---- backdrop blur (glcontext.nim:1743-1841)
    def draw_backdrop_blur(self, rect, radii_x, radii_y, blur_radius):
        x, y, w, h = [f32(v) for v in rect]
        if f32(blur_radius) <= 0 or w <= 0 or h <= 0:
            return
        g = _gl
        g.glActiveTexture(GL_TEXTURE0 + 2)
        g.glBindTexture(GL_TEXTURE_2D, self.backdrop_tex)
        g.glCopyTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, 0, 0, self.W, self.H)
        self._run_blur(f32(blur_radius))
        white = (255, 255, 255, 255)
        self.draw_rounded_rect_sdf(rect, [white] * 4, radii_x, radii_y, 17, blur_radius, 0.0)

    def _run_blur(self, radius):
        if radius <= f32(0.5):
            return
        g = _gl
        g.glDisable(GL_BLEND)
        g.glUseProgram(self.blur.id)
        for name, buf in (("vertexPos", self.blur_pos), ("vertexUv", self.blur_uv)):
            loc = self.blur.a(name)
            g.glBindBuffer(GL_ARRAY_BUFFER, buf)
            g.glEnableVertexAttribArray(loc)
            g.glVertexAttribPointer(loc, 2, GL_FLOAT, 0, 0, None)
        g.glUniform1i(self.blur.u("srcTex"), 0)
        g.glUniform1f(self.blur.u("blurRadius"), float(radius))
        g.glBindFramebuffer(GL_FRAMEBUFFER, self.blur_fbo)
        g.glViewport(0, 0, self.W, self.H)
        g.glActiveTexture(GL_TEXTURE0)
        g.glBindTexture(GL_TEXTURE_2D, self.backdrop_tex)
        g.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, self.backdrop_tmp, 0)
        g.glUniform2f(self.blur.u("texelStep"), float(f32(1) / f32(max(1, self.W))), 0.0)
        g.glDrawArrays(GL_TRIANGLES, 0, 3)
        g.glBindTexture(GL_TEXTURE_2D, self.backdrop_tmp)
        g.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, self.backdrop_tex, 0)
        g.glUniform2f(self.blur.u("texelStep"), 0.0, float(f32(1) / f32(max(1, self.H))))
        g.glDrawArrays(GL_TRIANGLES, 0, 3)
        g.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        for name in ("vertexPos", "vertexUv"):
            g.glDisableVertexAttribArray(self.blur.a(name))
        g.glEnable(GL_BLEND)

    def blur_only(self, rgba: np.ndarray, radius: float, passes="hv") -> np.ndarray:
        """Run blur.frag over an arbitrary top-down RGBA8 image (unit-test hook)."""
        g = _gl
        img = np.ascontiguousarray(rgba[::-1])  # GL rows are bottom-up
        g.glBindTexture(GL_TEXTURE_2D, self.backdrop_tex)
        g.glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, self.W, self.H, GL_RGBA, GL_UNSIGNED_BYTE, img.ctypes.data)
        if passes == "hv":
            self._run_blur(f32(radius))
            src = self.backdrop_tex
        else:
            raise ValueError(passes)
        g.glBindFramebuffer(GL_FRAMEBUFFER, self.blur_fbo)
        g.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, src, 0)
        out = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        g.glReadPixels(0, 0, self.W, self.H, GL_RGBA, GL_UNSIGNED_BYTE, out.ctypes.data)
        g.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        return out[::-1].copy()

    # ---- readback (glcontext.nim:2094-2135): RGBA8, flipped to top-down
    def read_pixels(self) -> np.ndarray:
        g = _gl
        g.glFinish()
        g.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        out = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        g.glReadPixels(0, 0, self.W, self.H, GL_RGBA, GL_UNSIGNED_BYTE, out.ctypes.data)
        return out[::-1].copy()

    def read_mask(self, level: int) -> np.ndarray:
        g = _gl
        g.glBindFramebuffer(GL_FRAMEBUFFER, self.mask_fbo)
        g.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, self.mask_texs[level], 0)
        out = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        g.glReadPixels(0, 0, self.W, self.H, GL_RGBA, GL_UNSIGNED_BYTE, out.ctypes.data)
        g.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        return out[::-1].copy()


def replay(calls, width, height, atlas_size=1024, images=None) -> np.ndarray:
    """Replay a recorded BackendContext call stream (list of [name, *args]) and
    return the top-down RGBA8 frame.  ``images`` maps key -> HxWx4 uint8 and is
    uploaded (in sorted-key order) before the frame begins."""
    gl = RefGL(width, height, atlas_size=atlas_size)
    try:
        for key in sorted(images or {}):
            gl.put_image(key, images[key])
        for call in calls:
            name, args = call[0], call[1:]
            getattr(gl, name)(*args)
        return gl.read_pixels()
    finally:
        gl.close()
