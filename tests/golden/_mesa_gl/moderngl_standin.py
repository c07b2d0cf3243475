"""A minimal stand-in for the ``moderngl`` PACKAGE (not for OpenGL) over a real OpenGL 3.3 core context
on Mesa's software rasteriser.  BUILD-CONTAINER TOOL of tests/golden/gen_golden_gl.py only.

moderngl 5.12 is a thin object wrapper around GL calls and is not installed here; OpenGL itself
(Mesa 23.2 llvmpipe) is.  This module exposes exactly the moderngl surface the reference's
``persp_proj`` touches (src/alproj/project.py:210-290) and forwards every call to the GL entry point
moderngl would issue for it, so that the reference's OWN shaders, matrices, buffers and draw call run
on a conformant GL:

    create_standalone_context, DEPTH_TEST, CULL_FACE
    Context.enable / buffer / program / vertex_array / renderbuffer / depth_renderbuffer /
            framebuffer / release
    Program[...] .value (mat4 as 16 floats taken column-major, transpose = GL_FALSE; float)
    VertexArray.render() (GL_TRIANGLES, all indices, 32-bit)
    Framebuffer.use() (bind + viewport = size) / clear(r, g, b, a) (colour + depth 1.0) /
            read(dtype='f4') (3 components, pack alignment 1)
    *.release()

State moderngl leaves at GL defaults stays at GL defaults: depth function GL_LESS (override with
``DEPTH_FUNC`` below to record what GL_LEQUAL would give -- moderngl's context default cannot be
checked without the package), cull GL_BACK, front face GL_CCW, renderbuffers RGBA32F and
DEPTH_COMPONENT24 (moderngl: ``renderbuffer(size, components=4, dtype)``, ``depth_renderbuffer(size)``).

Diagnostics the generator may ask for (they never change what the reference receives): after each
``VertexArray.render()`` the window-space depth buffer and a ``gl_PrimitiveID`` image of the same draw
are kept in ``LAST`` (the id pass re-draws the same vertex array, with the reference's vertex shader
as compiled, into a second framebuffer).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"

# moderngl's enable flags
BLEND, DEPTH_TEST, CULL_FACE = 1, 2, 4

DEPTH_FUNC = None          # None = leave GL's default (GL_LESS); or "<=" to force GL_LEQUAL
KEEP_DIAGNOSTICS = False
LAST = {}

GL_DEPTH_TEST, GL_CULL_FACE = 0x0B71, 0x0B44
GL_ARRAY_BUFFER, GL_ELEMENT_ARRAY_BUFFER, GL_STATIC_DRAW = 0x8892, 0x8893, 0x88E4
GL_VERTEX_SHADER, GL_FRAGMENT_SHADER = 0x8B31, 0x8B30
GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B81, 0x8B82
GL_FLOAT, GL_UNSIGNED_INT, GL_TRIANGLES = 0x1406, 0x1405, 0x0004
GL_RENDERBUFFER, GL_FRAMEBUFFER = 0x8D41, 0x8D40
GL_RGBA32F, GL_DEPTH_COMPONENT24 = 0x8814, 0x81A6
GL_COLOR_ATTACHMENT0, GL_DEPTH_ATTACHMENT = 0x8CE0, 0x8D00
GL_FRAMEBUFFER_COMPLETE = 0x8CD5
GL_COLOR_BUFFER_BIT, GL_DEPTH_BUFFER_BIT = 0x4000, 0x0100
GL_RGB, GL_RGBA, GL_DEPTH_COMPONENT = 0x1907, 0x1908, 0x1902
GL_PACK_ALIGNMENT = 0x0D05
GL_LEQUAL = 0x0203
GL_NO_ERROR = 0

_lib = None
_fn = {}

_SIGS = {
    "glEnable": (None, [ctypes.c_uint]),
    "glDepthFunc": (None, [ctypes.c_uint]),
    "glGetError": (ctypes.c_uint, []),
    "glGetString": (ctypes.c_char_p, [ctypes.c_uint]),
    "glGetIntegerv": (None, [ctypes.c_uint, ctypes.POINTER(ctypes.c_int)]),
    "glGenBuffers": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glDeleteBuffers": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glBindBuffer": (None, [ctypes.c_uint, ctypes.c_uint]),
    "glBufferData": (None, [ctypes.c_uint, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_uint]),
    "glCreateShader": (ctypes.c_uint, [ctypes.c_uint]),
    "glShaderSource": (None, [ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int)]),
    "glCompileShader": (None, [ctypes.c_uint]),
    "glGetShaderiv": (None, [ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_int)]),
    "glGetShaderInfoLog": (None, [ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p]),
    "glDeleteShader": (None, [ctypes.c_uint]),
    "glCreateProgram": (ctypes.c_uint, []),
    "glAttachShader": (None, [ctypes.c_uint, ctypes.c_uint]),
    "glLinkProgram": (None, [ctypes.c_uint]),
    "glGetProgramiv": (None, [ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_int)]),
    "glGetProgramInfoLog": (None, [ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p]),
    "glUseProgram": (None, [ctypes.c_uint]),
    "glDeleteProgram": (None, [ctypes.c_uint]),
    "glGetUniformLocation": (ctypes.c_int, [ctypes.c_uint, ctypes.c_char_p]),
    "glGetAttribLocation": (ctypes.c_int, [ctypes.c_uint, ctypes.c_char_p]),
    "glUniformMatrix4fv": (None, [ctypes.c_int, ctypes.c_int, ctypes.c_ubyte, ctypes.POINTER(ctypes.c_float)]),
    "glUniform1f": (None, [ctypes.c_int, ctypes.c_float]),
    "glGenVertexArrays": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glDeleteVertexArrays": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glBindVertexArray": (None, [ctypes.c_uint]),
    "glEnableVertexAttribArray": (None, [ctypes.c_uint]),
    "glVertexAttribPointer": (None, [ctypes.c_uint, ctypes.c_int, ctypes.c_uint, ctypes.c_ubyte, ctypes.c_int, ctypes.c_void_p]),
    "glDrawElements": (None, [ctypes.c_uint, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]),
    "glGenRenderbuffers": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glDeleteRenderbuffers": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glBindRenderbuffer": (None, [ctypes.c_uint, ctypes.c_uint]),
    "glRenderbufferStorage": (None, [ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int]),
    "glGenFramebuffers": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glDeleteFramebuffers": (None, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]),
    "glBindFramebuffer": (None, [ctypes.c_uint, ctypes.c_uint]),
    "glFramebufferRenderbuffer": (None, [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]),
    "glCheckFramebufferStatus": (ctypes.c_uint, [ctypes.c_uint]),
    "glViewport": (None, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "glClearColor": (None, [ctypes.c_float] * 4),
    "glClearDepth": (None, [ctypes.c_double]),
    "glClear": (None, [ctypes.c_uint]),
    "glPixelStorei": (None, [ctypes.c_uint, ctypes.c_int]),
    "glReadPixels": (None, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]),
    "glFinish": (None, []),
}


def _load():
    global _lib
    if _lib is not None:
        return
    so = os.path.join(_HERE, "libdri_ctx.so")
    src = os.path.join(_HERE, "dri_ctx.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", src, "-o", so, "-ldl"], check=True)
    _lib = ctypes.CDLL(so)
    _lib.dri_ctx_error.restype = ctypes.c_char_p
    _lib.dri_ctx_proc.restype = ctypes.c_void_p
    if _lib.dri_ctx_create(DRIVER.encode(), 3, 3) != 0:
        raise RuntimeError("no headless GL context: " + _lib.dri_ctx_error().decode())
    for name, (res, args) in _SIGS.items():
        addr = _lib.dri_ctx_proc(name.encode())
        if not addr:
            raise RuntimeError(f"GL entry point {name} missing")
        _fn[name] = ctypes.CFUNCTYPE(res, *args)(addr)


def _gl(name, *args):
    r = _fn[name](*args)
    e = _fn["glGetError"]()
    if e != GL_NO_ERROR:
        raise RuntimeError(f"{name}: GL error 0x{e:04x}")
    return r


def _gen(name):
    v = ctypes.c_uint(0)
    _gl(name, 1, ctypes.byref(v))
    return v.value


def gl_info():
    _load()
    sub = ctypes.c_int(0)
    _gl("glGetIntegerv", 0x0D50, ctypes.byref(sub))
    return dict(vendor=_gl("glGetString", 0x1F00).decode(), renderer=_gl("glGetString", 0x1F01).decode(),
                version=_gl("glGetString", 0x1F02).decode(), subpixel_bits=sub.value)


class Buffer:
    def __init__(self, data):
        self.glo = _gen("glGenBuffers")
        self.size = len(data)
        _gl("glBindBuffer", GL_ARRAY_BUFFER, self.glo)
        _gl("glBufferData", GL_ARRAY_BUFFER, len(data), ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p), GL_STATIC_DRAW)

    def release(self):
        _gl("glDeleteBuffers", 1, ctypes.byref(ctypes.c_uint(self.glo)))


class _Uniform:
    def __init__(self, prog, name):
        self._prog, self._loc = prog, _gl("glGetUniformLocation", prog.glo, name.encode())
        if self._loc < 0:
            raise KeyError(name)

    @property
    def value(self):
        raise NotImplementedError

    @value.setter
    def value(self, v):
        _gl("glUseProgram", self._prog.glo)
        if isinstance(v, (tuple, list)) and len(v) == 16:
            arr = (ctypes.c_float * 16)(*[float(x) for x in v])
            _gl("glUniformMatrix4fv", self._loc, 1, 0, arr)       # moderngl: transpose = GL_FALSE
        else:
            _gl("glUniform1f", self._loc, float(v))


def _compile(kind, text):
    sh = _gl("glCreateShader", kind)
    src = ctypes.c_char_p(text.encode())
    _gl("glShaderSource", sh, 1, ctypes.byref(src), None)
    _gl("glCompileShader", sh)
    ok = ctypes.c_int(0)
    _gl("glGetShaderiv", sh, GL_COMPILE_STATUS, ctypes.byref(ok))
    if not ok.value:
        log = ctypes.create_string_buffer(4096)
        _gl("glGetShaderInfoLog", sh, 4096, None, log)
        raise RuntimeError("shader compile: " + log.value.decode())
    return sh


class Program:
    def __init__(self, vertex_shader, fragment_shader):
        self._vs_text = vertex_shader
        vs, fs = _compile(GL_VERTEX_SHADER, vertex_shader), _compile(GL_FRAGMENT_SHADER, fragment_shader)
        self.glo = _gl("glCreateProgram")
        _gl("glAttachShader", self.glo, vs)
        _gl("glAttachShader", self.glo, fs)
        _gl("glLinkProgram", self.glo)
        ok = ctypes.c_int(0)
        _gl("glGetProgramiv", self.glo, GL_LINK_STATUS, ctypes.byref(ok))
        if not ok.value:
            log = ctypes.create_string_buffer(4096)
            _gl("glGetProgramInfoLog", self.glo, 4096, None, log)
            raise RuntimeError("program link: " + log.value.decode())
        _gl("glDeleteShader", vs)
        _gl("glDeleteShader", fs)
        self._uniforms = {}
        self._values = {}

    def __getitem__(self, name):
        if name not in self._uniforms:
            self._uniforms[name] = _Uniform(self, name)
        u = self._uniforms[name]
        prog = self

        class _Rec:
            @property
            def value(self_inner):
                return prog._values.get(name)

            @value.setter
            def value(self_inner, v):
                prog._values[name] = v
                u.value = v
        return _Rec()

    def release(self):
        _gl("glDeleteProgram", self.glo)


_FORMATS = {"3f": (3, GL_FLOAT, 12), "2f": (2, GL_FLOAT, 8), "1f": (1, GL_FLOAT, 4), "4f": (4, GL_FLOAT, 16)}


class VertexArray:
    def __init__(self, program, content, index_buffer):
        self.program, self.index_buffer, self.content = program, index_buffer, content
        self.glo = _gen("glGenVertexArrays")
        self._bind_attribs(program)
        self.n_indices = index_buffer.size // 4

    def _bind_attribs(self, program):
        _gl("glBindVertexArray", self.glo)
        for buf, fmt, name in self.content:
            n, typ, stride = _FORMATS[fmt]
            loc = _gl("glGetAttribLocation", program.glo, name.encode())
            if loc < 0:
                continue                       # moderngl skips attributes the linker removed
            _gl("glBindBuffer", GL_ARRAY_BUFFER, buf.glo)
            _gl("glEnableVertexAttribArray", loc)
            _gl("glVertexAttribPointer", loc, n, typ, 0, stride, None)
        _gl("glBindBuffer", GL_ELEMENT_ARRAY_BUFFER, self.index_buffer.glo)

    def render(self):
        _gl("glUseProgram", self.program.glo)
        _gl("glBindVertexArray", self.glo)
        _gl("glDrawElements", GL_TRIANGLES, self.n_indices, GL_UNSIGNED_INT, None)
        if KEEP_DIAGNOSTICS:
            self._diagnostics()

    def _diagnostics(self):
        fbo = Framebuffer.current
        w, h = fbo.size
        _gl("glFinish")
        depth = np.empty((h, w), dtype=np.float32)
        _gl("glPixelStorei", GL_PACK_ALIGNMENT, 1)
        _gl("glReadPixels", 0, 0, w, h, GL_DEPTH_COMPONENT, GL_FLOAT, depth.ctypes.data_as(ctypes.c_void_p))
        LAST["depth"] = depth
        # gl_PrimitiveID of the same draw: the reference's vertex shader (as handed to Program), a fragment
        # shader of ours that writes the id, a second framebuffer with its own depth buffer, same state
        idp = Program(self.program._vs_text, """
            #version 330
            in vec3 v_color;
            in float v_distance;
            layout(location=0) out vec4 f_color;
            // the id in two exactly representable halves: one float32 holds integers up to 2^24 only, and the c4-sized
            // draw has 72 M triangles
            void main() { int id = gl_PrimitiveID + 1; f_color = vec4(float(id & 0xFFFFF), v_distance, float(id >> 20), 1.0); }
        """)
        for k in ("proj", "view"):
            idp[k].value = self.program._values[k]
        rb, db = Renderbuffer((w, h), GL_RGBA32F), Renderbuffer((w, h), GL_DEPTH_COMPONENT24)
        fb2 = Framebuffer(rb, db)
        fb2.use()
        fb2.clear(0.0, 0.0, 0.0, 1.0)
        va2 = VertexArray(idp, self.content, self.index_buffer)
        _gl("glUseProgram", idp.glo)
        _gl("glBindVertexArray", va2.glo)
        _gl("glDrawElements", GL_TRIANGLES, self.n_indices, GL_UNSIGNED_INT, None)
        img = np.frombuffer(fb2.read(dtype="f4"), dtype=np.float32).reshape(h, w, 3)
        LAST["prim_id"] = (np.rint(img[:, :, 0]).astype(np.int64) | (np.rint(img[:, :, 2]).astype(np.int64) << 20)) - 1      # -1 = nothing drawn
        LAST["frag_distance"] = img[:, :, 1].copy()
        va2.release(), fb2.release(), rb.release(), db.release(), idp.release()
        fbo.use()
        _gl("glBindVertexArray", self.glo)

    def release(self):
        _gl("glDeleteVertexArrays", 1, ctypes.byref(ctypes.c_uint(self.glo)))


class Renderbuffer:
    def __init__(self, size, internal):
        self.size = tuple(int(s) for s in size)
        self.glo = _gen("glGenRenderbuffers")
        _gl("glBindRenderbuffer", GL_RENDERBUFFER, self.glo)
        _gl("glRenderbufferStorage", GL_RENDERBUFFER, internal, self.size[0], self.size[1])

    def release(self):
        _gl("glDeleteRenderbuffers", 1, ctypes.byref(ctypes.c_uint(self.glo)))


class Framebuffer:
    current = None

    def __init__(self, color, depth):
        self.size = color.size
        self.glo = _gen("glGenFramebuffers")
        _gl("glBindFramebuffer", GL_FRAMEBUFFER, self.glo)
        _gl("glFramebufferRenderbuffer", GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, color.glo)
        _gl("glFramebufferRenderbuffer", GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, depth.glo)
        st = _gl("glCheckFramebufferStatus", GL_FRAMEBUFFER)
        if st != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError(f"framebuffer incomplete: 0x{st:04x}")
        prev = Framebuffer.current
        _gl("glBindFramebuffer", GL_FRAMEBUFFER, prev.glo if prev is not None else 0)

    def use(self):
        _gl("glBindFramebuffer", GL_FRAMEBUFFER, self.glo)
        _gl("glViewport", 0, 0, self.size[0], self.size[1])
        Framebuffer.current = self

    def clear(self, red=0.0, green=0.0, blue=0.0, alpha=0.0, depth=1.0):
        _gl("glBindFramebuffer", GL_FRAMEBUFFER, self.glo)
        _gl("glClearColor", red, green, blue, alpha)
        _gl("glClearDepth", depth)
        _gl("glClear", GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT)
        cur = Framebuffer.current
        _gl("glBindFramebuffer", GL_FRAMEBUFFER, cur.glo if cur is not None else 0)

    def read(self, dtype="f1", components=3):
        assert dtype == "f4"
        w, h = self.size
        out = np.empty(h * w * components, dtype=np.float32)
        _gl("glBindFramebuffer", GL_FRAMEBUFFER, self.glo)
        _gl("glPixelStorei", GL_PACK_ALIGNMENT, 1)
        _gl("glReadPixels", 0, 0, w, h, GL_RGB if components == 3 else GL_RGBA, GL_FLOAT, out.ctypes.data_as(ctypes.c_void_p))
        cur = Framebuffer.current
        _gl("glBindFramebuffer", GL_FRAMEBUFFER, cur.glo if cur is not None else 0)
        return out.tobytes()

    def release(self):
        if Framebuffer.current is self:
            Framebuffer.current = None
            _gl("glBindFramebuffer", GL_FRAMEBUFFER, 0)
        _gl("glDeleteFramebuffers", 1, ctypes.byref(ctypes.c_uint(self.glo)))


class Context:
    """One process-wide GL context (the reference creates and releases one per call; releasing ours is a
    no-op so that the next call can reuse it -- GL state set per call is re-established by enable())."""

    def enable(self, flags):
        if flags & DEPTH_TEST:
            _gl("glEnable", GL_DEPTH_TEST)
            if DEPTH_FUNC == "<=":
                _gl("glDepthFunc", GL_LEQUAL)
            else:
                _gl("glDepthFunc", 0x0201)      # GL_LESS, the GL default (restored in case a previous call changed it)
        if flags & CULL_FACE:
            _gl("glEnable", GL_CULL_FACE)

    def buffer(self, data):
        return Buffer(bytes(data))

    def program(self, vertex_shader, fragment_shader):
        return Program(vertex_shader, fragment_shader)

    def vertex_array(self, program, content, index_buffer):
        return VertexArray(program, content, index_buffer)

    def renderbuffer(self, size, components=4, dtype="f1"):
        assert components == 4 and dtype == "f4"
        return Renderbuffer(size, GL_RGBA32F)

    def depth_renderbuffer(self, size):
        return Renderbuffer(size, GL_DEPTH_COMPONENT24)

    def framebuffer(self, color, depth):
        return Framebuffer(color, depth)

    def release(self):
        pass


def create_standalone_context():
    _load()
    return Context()
