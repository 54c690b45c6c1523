/*
 * ref_gl_runner.c -- TEST INFRASTRUCTURE, AUTHORING CONTAINER ONLY.
 *
 * Runs the reference's UNMODIFIED compute shader (/root/reference/VolumeRenderer.cs,
 * read from where it lies at run time, never copied) under a real OpenGL
 * implementation: Mesa 23.2.1 llvmpipe, loaded straight from
 * /usr/lib/x86_64-linux-gnu/dri/swrast_dri.so through the DRI swrast interface
 * (GL/internal/dri_interface.h).  There is no X server, EGL or OSMesa in the image,
 * so this file is the ~loader a GLX/EGL front end would be: it hands the driver a
 * DRI_SWRastLoader with a dummy 16x16 drawable, asks for a GL 4.3 core context (what
 * /root/reference/src/GlfwManager.cpp:43-47 requests), and fetches entry points from
 * libglapi.  Nothing here stands in for the GL: GLSL compiler, texture sampling,
 * image stores, UBO layout and dispatch are all Mesa's.
 *
 * After the context is current it issues exactly the reference's GL sequence:
 *   setupFBO          src/RendererCore.cpp:184-219   RGBA32F target texture
 *   setupUBO          src/RendererCore.cpp:221-240   21 floats, binding 1
 *   readVolumeData    src/RendererCore.cpp:408-419   R8UI/R16UI 3-D texture, unit 1,
 *                                                    CLAMP_TO_EDGE, MIN/MAG filter
 *   createShader / createShaderProgram               src/RendererCore.cpp:449-538
 *   loadShader        src/RendererCore.cpp:119-125   workgroups = window / local_size
 *   setUniforms       src/RendererCore.cpp:56-110    loc 0..6
 *   render            src/RendererCore.cpp:138-163   bind image, dispatch, barrier
 * and reads the RGBA32F texture back (glGetTexImage).
 *
 * `tex_filter` is the MIN/MAG filter put on the volume texture: GL_LINEAR (0x2601) is
 * what the reference sets; GL_NEAREST (0x2600) is the single stated deviation used to
 * measure SURVEY F4.  Mesa's driconf switch `force_integer_tex_nearest=true` (env var,
 * read by the driver at screen creation) is the third way to run it: reference calls
 * untouched, integer-texture completeness relaxed the way the vendor drivers do.
 *
 * Built by oracle/Makefile into oracle/_ref/libref_gl.so (git-ignored).  Only
 * oracle/ref_gl/mint_ref_gl_goldens.py calls it; the frames it writes under
 * tests/golden/ are the data that travels.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

#define DRIVER_PATH "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"

/* ---- DRI_SWRastLoader: a drawable nobody looks at ------------------------------- */
static void ld_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = *y = 0; *w = *h = 16; }
static void ld_put(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_get(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void ld_put2(__DRIdrawable *d, int op, int x, int y, int w, int h, int s, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)s; (void)data; (void)p; }
static void ld_get2(__DRIdrawable *d, int x, int y, int w, int h, int s, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)s * h); }

static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ld_info, .putImage = ld_put, .getImage = ld_get,
    .putImage2 = ld_put2, .getImage2 = ld_get2,
};
static const __DRIextension *g_loader_ext[] = { &g_loader.base, NULL };

/* ---- GL entry points (libglapi dispatch stubs) ---------------------------------- */
#define GLFUNCS(X) \
    X(PFNGLGETSTRINGPROC, GetString) X(PFNGLGETERRORPROC, GetError) \
    X(PFNGLGENFRAMEBUFFERSPROC, GenFramebuffers) X(PFNGLBINDFRAMEBUFFERPROC, BindFramebuffer) \
    X(PFNGLGENTEXTURESPROC, GenTextures) X(PFNGLACTIVETEXTUREPROC, ActiveTexture) \
    X(PFNGLBINDTEXTUREPROC, BindTexture) X(PFNGLTEXIMAGE2DPROC, TexImage2D) \
    X(PFNGLTEXPARAMETERIPROC, TexParameteri) X(PFNGLFRAMEBUFFERTEXTURE2DPROC, FramebufferTexture2D) \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, CheckFramebufferStatus) X(PFNGLREADBUFFERPROC, ReadBuffer) \
    X(PFNGLGENBUFFERSPROC, GenBuffers) X(PFNGLBINDBUFFERPROC, BindBuffer) \
    X(PFNGLBUFFERDATAPROC, BufferData) X(PFNGLBINDBUFFERBASEPROC, BindBufferBase) \
    X(PFNGLPIXELSTOREIPROC, PixelStorei) X(PFNGLTEXIMAGE3DPROC, TexImage3D) \
    X(PFNGLCREATESHADERPROC, CreateShader) X(PFNGLSHADERSOURCEPROC, ShaderSource) \
    X(PFNGLCOMPILESHADERPROC, CompileShader) X(PFNGLGETSHADERIVPROC, GetShaderiv) \
    X(PFNGLGETSHADERINFOLOGPROC, GetShaderInfoLog) X(PFNGLDELETESHADERPROC, DeleteShader) \
    X(PFNGLCREATEPROGRAMPROC, CreateProgram) X(PFNGLATTACHSHADERPROC, AttachShader) \
    X(PFNGLLINKPROGRAMPROC, LinkProgram) X(PFNGLDETACHSHADERPROC, DetachShader) \
    X(PFNGLGETPROGRAMIVPROC, GetProgramiv) X(PFNGLGETPROGRAMINFOLOGPROC, GetProgramInfoLog) \
    X(PFNGLDELETEPROGRAMPROC, DeleteProgram) X(PFNGLUSEPROGRAMPROC, UseProgram) \
    X(PFNGLUNIFORM1FPROC, Uniform1f) X(PFNGLUNIFORM3FPROC, Uniform3f) X(PFNGLUNIFORM1IPROC, Uniform1i) \
    X(PFNGLBINDIMAGETEXTUREPROC, BindImageTexture) X(PFNGLDISPATCHCOMPUTEPROC, DispatchCompute) \
    X(PFNGLMEMORYBARRIERPROC, MemoryBarrier) X(PFNGLGETTEXIMAGEPROC, GetTexImage) \
    X(PFNGLFINISHPROC, Finish) X(PFNGLDELETETEXTURESPROC, DeleteTextures) \
    X(PFNGLDELETEBUFFERSPROC, DeleteBuffers) X(PFNGLDELETEFRAMEBUFFERSPROC, DeleteFramebuffers) \
    X(PFNGLCLEARTEXIMAGEPROC, ClearTexImage) X(PFNGLGETBUFFERSUBDATAPROC, GetBufferSubData)
#define X(T, n) static T gl##n;
GLFUNCS(X)
#undef X

static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_sw;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static char g_err[4096];

static int fail(const char *fmt, const char *a)
{
    snprintf(g_err, sizeof g_err, fmt, a ? a : "");
    return -1;
}

const char *refgl_last_error(void) { return g_err; }

/* Brings up the llvmpipe GL 4.3 core context.  info (optional) receives
   "GL_VERSION | GL_RENDERER | GLSL version". */
int refgl_init(char *info, int info_len)
{
    if (g_ctx) goto describe;
    void *h = dlopen(DRIVER_PATH, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail("dlopen: %s", dlerror());
    const __DRIextension **(*get_ext)(void) =
        (const __DRIextension **(*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
    if (!get_ext) return fail("%s", "__driDriverGetExtensions_swrast not exported");
    const __DRIextension **ext = get_ext();
    for (int i = 0; ext[i]; i++) {
        if (!strcmp(ext[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)ext[i];
        if (!strcmp(ext[i]->name, __DRI_SWRAST)) g_sw = (const __DRIswrastExtension *)ext[i];
    }
    if (!g_core || !g_sw || g_sw->base.version < 4) return fail("%s", "DRI_Core / DRI_SWRast v4 missing");
    const __DRIconfig **configs = NULL;
    g_screen = g_sw->createNewScreen2(0, g_loader_ext, ext, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) return fail("%s", "createNewScreen2 failed");
    /* src/GlfwManager.cpp:43-47: OpenGL 4.3, core profile */
    const uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3 };
    unsigned err = 0;
    g_ctx = g_sw->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!g_ctx) return fail("%s", "createContextAttribs(GL 4.3 core) failed");
    g_draw = g_sw->createNewDrawable(g_screen, configs[0], NULL);
    if (!g_draw) return fail("%s", "createNewDrawable failed");
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) return fail("%s", "bindContext failed");
    void *(*gpa)(const char *) = (void *(*)(const char *))dlsym(RTLD_DEFAULT, "_glapi_get_proc_address");
    if (!gpa) return fail("%s", "_glapi_get_proc_address not found");
#define X(T, n) gl##n = (T)gpa("gl" #n); if (!gl##n) return fail("missing entry point %s", "gl" #n);
    GLFUNCS(X)
#undef X
describe:
    if (info && info_len > 0)
        snprintf(info, (size_t)info_len, "%s | %s | GLSL %s", (const char *)glGetString(GL_VERSION),
                 (const char *)glGetString(GL_RENDERER), (const char *)glGetString(GL_SHADING_LANGUAGE_VERSION));
    return 0;
}

typedef struct refgl_job {
    const char *shader_path;       /* /root/reference/VolumeRenderer.cs, as it lies */
    const void *volume;            /* x fastest, then y, then z */
    int32_t nx, ny, nz, bytes_per_voxel;
    int32_t fb_w, fb_h;            /* framebuffer_size */
    int32_t win_w, win_h;          /* window_size (workgroups = window / 16) */
    float cam[21];                 /* Camera::setUBO contents */
    float alpha_scale;             /* loc 0 */
    float voxel_size[3];           /* loc 1 */
    int32_t min_val, max_val;      /* loc 2/3 AS UPLOADED (after the +1000 for 16-bit) */
    int32_t is_mip, view_top, view_bottom;   /* loc 4/5/6 */
    int32_t tex_filter;            /* GL_LINEAR = reference; GL_NEAREST = stated deviation */
    float clear_value;             /* target is pre-filled with this (to see untouched pixels) */
    int32_t tex_float;             /* 0 = the reference's R8UI / R16UI upload; 1 = `volume` holds floats, uploaded as GL_R32F (the
                                      TRILINEAR cross-check only: a filterable texture for a shader with `sampler3D`) */
} refgl_job;

static int slurp(const char *path, char **out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *s = (char *)malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(s); return -1; }
    s[n] = 0;
    fclose(f);
    *out = s;
    return 0;
}

/* One frame.  out_rgba: fb_h*fb_w*4 floats, row 0 = bottom (GL origin). */
int refgl_render(const refgl_job *j, float *out_rgba)
{
    if (!g_ctx && refgl_init(NULL, 0)) return -1;
    int rc = -1;
    GLuint fbo = 0, fbo_tex = 0, vol_tex = 0, ubo = 0, cs = 0, prog = 0;
    char *src = NULL;
    while (glGetError() != GL_NO_ERROR) {}

    /* setupFBO (RendererCore.cpp:184-219) */
    glGenFramebuffers(1, &fbo);
    glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    glGenTextures(1, &fbo_tex);
    glActiveTexture(GL_TEXTURE0);
    glBindTexture(GL_TEXTURE_2D, fbo_tex);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, j->fb_w, j->fb_h, 0, GL_RGBA, GL_FLOAT, 0);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    glBindTexture(GL_TEXTURE_2D, 0);
    glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, fbo_tex, 0);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fail("%s", "framebuffer incomplete"); goto out; }
    {   /* not a reference call: make pixels the dispatch never reaches (Q1) visible */
        const float c[4] = { j->clear_value, j->clear_value, j->clear_value, j->clear_value };
        glClearTexImage(fbo_tex, 0, GL_RGBA, GL_FLOAT, c);
    }
    /* setup (RendererCore.cpp:34-44) */
    glBindFramebuffer(GL_READ_FRAMEBUFFER, fbo);
    glReadBuffer(GL_COLOR_ATTACHMENT0);
    glBindFramebuffer(GL_DRAW_FRAMEBUFFER, 0);
    glGenTextures(1, &vol_tex);

    /* setupUBO (RendererCore.cpp:221-240) */
    glGenBuffers(1, &ubo);
    glBindBuffer(GL_UNIFORM_BUFFER, ubo);
    glBufferData(GL_UNIFORM_BUFFER, sizeof(float) * 21, j->cam, GL_DYNAMIC_DRAW);
    glBindBufferBase(GL_UNIFORM_BUFFER, 1, ubo);
    glBindBuffer(GL_UNIFORM_BUFFER, 0);

    /* readVolumeData, upload part (RendererCore.cpp:407-419,433) */
    glActiveTexture(GL_TEXTURE1);
    glBindTexture(GL_TEXTURE_3D, vol_tex);
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_WRAP_R, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MAG_FILTER, j->tex_filter);
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MIN_FILTER, j->tex_filter);
    if (j->nx % 4 != 0) glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    if (j->tex_float)
        glTexImage3D(GL_TEXTURE_3D, 0, GL_R32F, j->nx, j->ny, j->nz, 0, GL_RED, GL_FLOAT, j->volume);
    else
        glTexImage3D(GL_TEXTURE_3D, 0, j->bytes_per_voxel == 1 ? GL_R8UI : GL_R16UI, j->nx, j->ny, j->nz, 0,
                     GL_RED_INTEGER, j->bytes_per_voxel == 1 ? GL_UNSIGNED_BYTE : GL_UNSIGNED_SHORT, j->volume);
    glPixelStorei(GL_UNPACK_ALIGNMENT, 4);
    if (glGetError() != GL_NO_ERROR) { fail("%s", "GL error during texture / UBO set-up"); goto out; }

    /* createShader (RendererCore.cpp:449-503): the file, byte for byte */
    if (slurp(j->shader_path, &src)) { fail("cannot read shader %s", j->shader_path); goto out; }
    cs = glCreateShader(GL_COMPUTE_SHADER);
    { const GLchar *s = src; glShaderSource(cs, 1, &s, 0); }
    glCompileShader(cs);
    GLint ok = 0;
    glGetShaderiv(cs, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[3500] = { 0 };
        glGetShaderInfoLog(cs, sizeof log - 1, NULL, log);
        fail("shader compile failed:\n%s", log);
        goto out;
    }
    /* createShaderProgram (RendererCore.cpp:505-538) */
    prog = glCreateProgram();
    glAttachShader(prog, cs);
    glLinkProgram(prog);
    glDetachShader(prog, cs);
    glDeleteShader(cs);
    cs = 0;
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[3500] = { 0 };
        glGetProgramInfoLog(prog, sizeof log - 1, NULL, log);
        fail("program link failed:\n%s", log);
        goto out;
    }
    /* loadShader (RendererCore.cpp:119-125) */
    GLint wg[3] = { 0, 0, 0 };
    glGetProgramiv(prog, GL_COMPUTE_WORK_GROUP_SIZE, wg);
    const GLuint groups_x = (GLuint)(j->win_w / wg[0]), groups_y = (GLuint)(j->win_h / wg[1]);
    glUseProgram(prog);
    /* setUniforms (RendererCore.cpp:56-110) */
    glUniform1f(0, j->alpha_scale);
    glUniform3f(1, j->voxel_size[0], j->voxel_size[1], j->voxel_size[2]);
    glUniform1i(2, j->min_val);
    glUniform1i(3, j->max_val);
    glUniform1i(4, j->is_mip);
    glUniform1i(5, j->view_top);
    glUniform1i(6, j->view_bottom);
    if (glGetError() != GL_NO_ERROR) { fail("%s", "GL error while setting uniforms"); goto out; }

    /* render (RendererCore.cpp:138-163), minus the timer query and the blit */
    glBindImageTexture(0, fbo_tex, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_RGBA32F);
    if (groups_x && groups_y) glDispatchCompute(groups_x, groups_y, 1);
    glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
    glBindImageTexture(0, 0, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_RGBA32F);
    glFinish();

    glActiveTexture(GL_TEXTURE0);
    glBindTexture(GL_TEXTURE_2D, fbo_tex);
    glPixelStorei(GL_PACK_ALIGNMENT, 4);
    glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, out_rgba);
    glBindTexture(GL_TEXTURE_2D, 0);
    if (glGetError() != GL_NO_ERROR) { fail("%s", "GL error during dispatch / read-back"); goto out; }
    rc = 0;
out:
    free(src);
    if (cs) glDeleteShader(cs);
    glUseProgram(0);
    if (prog) glDeleteProgram(prog);
    if (vol_tex) glDeleteTextures(1, &vol_tex);
    if (fbo_tex) glDeleteTextures(1, &fbo_tex);
    if (ubo) glDeleteBuffers(1, &ubo);
    glBindFramebuffer(GL_FRAMEBUFFER, 0);
    if (fbo) glDeleteFramebuffers(1, &fbo);
    return rc;
}

/* Arithmetic probe: runs `src` (a compute shader with `buffer` blocks at binding 0 = in,
   1 = out) over `groups` workgroups and copies the output block back.  Used by
   oracle/ref_gl/probe_arith.py to MEASURE how this GL rounds a/b, normalize(), length(),
   mat*vec and a*b+c, which is what the oracle's VRO_ARITH_MESA mode restates. */
int refgl_compute(const char *src, const void *in, int in_bytes, void *out, int out_bytes, int groups)
{
    if (!g_ctx && refgl_init(NULL, 0)) return -1;
    int rc = -1;
    GLuint cs = 0, prog = 0, buf[2] = { 0, 0 };
    while (glGetError() != GL_NO_ERROR) {}
    cs = glCreateShader(GL_COMPUTE_SHADER);
    glShaderSource(cs, 1, &src, 0);
    glCompileShader(cs);
    GLint ok = 0;
    glGetShaderiv(cs, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[3500] = { 0 };
        glGetShaderInfoLog(cs, sizeof log - 1, NULL, log);
        fail("probe compile failed:\n%s", log);
        goto out;
    }
    prog = glCreateProgram();
    glAttachShader(prog, cs);
    glLinkProgram(prog);
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { fail("%s", "probe link failed"); goto out; }
    glGenBuffers(2, buf);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, buf[0]);
    glBufferData(GL_SHADER_STORAGE_BUFFER, in_bytes, in, GL_STATIC_DRAW);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, buf[1]);
    glBufferData(GL_SHADER_STORAGE_BUFFER, out_bytes, NULL, GL_DYNAMIC_READ);
    glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, buf[0]);
    glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, buf[1]);
    glUseProgram(prog);
    glDispatchCompute((GLuint)groups, 1, 1);
    glMemoryBarrier(GL_ALL_BARRIER_BITS);
    glFinish();
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, buf[1]);
    glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, out_bytes, out);
    if (glGetError() != GL_NO_ERROR) { fail("%s", "GL error in probe"); goto out; }
    rc = 0;
out:
    glUseProgram(0);
    if (cs) glDeleteShader(cs);
    if (prog) glDeleteProgram(prog);
    if (buf[0]) glDeleteBuffers(2, buf);
    return rc;
}
