// Declarations of the part of Pangolin's (and GL's) API that the reference's callers and headers use (syntax check only).
#pragma once
#include <unistd.h>
#include <cstddef>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>
typedef unsigned int GLuint;
typedef int GLint;
typedef int GLsizei;
typedef unsigned int GLenum;
typedef float GLfloat;
typedef void GLvoid;
typedef long GLsizeiptr;
typedef double GLdouble;
#define GL_LINE_STRIP 0x0003
#define GL_POINTS 0x0000
#define GL_FLOAT 0x1406
#define GL_UNSIGNED_BYTE 0x1401
#define GL_UNSIGNED_SHORT 0x1403
#define GL_RGB 0x1907
#define GL_RGBA 0x1908
#define GL_RED 0x1903
#define GL_RED_INTEGER 0x8D94
#define GL_RGB32F 0x8815
#define GL_RGBA32F 0x8814
#define GL_R32F 0x822E
#define GL_R16UI 0x8234
#define GL_R8UI 0x8232
#define GL_ARRAY_BUFFER 0x8892
#define GL_STREAM_DRAW 0x88E0
#define GL_MODELVIEW 0x1700
#define GL_PROJECTION 0x1701
#define GL_VERTEX_ARRAY 0x8074
inline void glColor3f(float, float, float) {}
// the fixed-function / buffer calls the facade's SM_FACADE_GL branches make (declarations: tests/test_reference_callers_compile.py
// type-checks those branches; nothing is linked)
void glGenBuffers(GLsizei, GLuint *);
void glBindBuffer(GLenum, GLuint);
void glBufferData(GLenum, GLsizeiptr, const void *, GLenum);
void glMatrixMode(GLenum);
void glLoadIdentity();
void glMultMatrixd(const GLdouble *);
void glMultMatrixf(const GLfloat *);
void glEnableClientState(GLenum);
void glDisableClientState(GLenum);
void glVertexPointer(GLint, GLenum, GLsizei, const void *);
void glDrawArrays(GLenum, GLint, GLsizei);
#define CheckGlDieOnError() ((void)0)
namespace pangolin {
struct GlTexture {
    GLint internal_format = 0;
    GLuint tid = 0;
    GLint width = 0, height = 0;
    // pangolin/gl/gl.h: Reinitialise(width, height, internal_format, sampling_linear, border, glformat, gltype, data = 0),
    // Upload(image, data_format, data_type), Upload(image, tex_x_offset, tex_y_offset, data_w, data_h, data_format, data_type)
    void Reinitialise(GLsizei width, GLsizei height, GLint internal_format, bool sampling_linear, int border, GLenum glformat, GLenum gltype,
                      GLvoid *data = 0);
    void Upload(const void *image, GLenum data_format, GLenum data_type);
    void Upload(const void *image, GLsizei tex_x_offset, GLsizei tex_y_offset, GLsizei data_w, GLsizei data_h, GLenum data_format, GLenum data_type);
};
struct GlFramebuffer {};
struct GlRenderBuffer {};
struct OpenGlMatrix {
    double m[16];
    OpenGlMatrix Inverse() const { return *this; }
    double operator()(int r, int c) const { return m[c * 4 + r]; }
};
struct OpenGlRenderState {
    void SetModelViewMatrix(const OpenGlMatrix &) {}
    OpenGlMatrix &GetModelViewMatrix() { return mv_; }
    OpenGlMatrix GetProjectionModelViewMatrix() const { return mv_; }
    OpenGlMatrix mv_;
};
template <class T> struct Var {
    const T &Get() const { return v_; }
    T v_;
};
inline bool Pushed(Var<bool> &) { return false; }
template <class C> inline void glDrawVertices(const C &, GLenum) {}
}  // namespace pangolin
