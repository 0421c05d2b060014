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
#define GL_LINE_STRIP 0x0003
#define GL_POINTS 0x0000
inline void glColor3f(float, float, float) {}
#define CheckGlDieOnError() ((void)0)
namespace pangolin {
struct GlTexture {
    GLint internal_format = 0;
    GLuint tid = 0;
    GLint width = 0, height = 0;
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
