// sm_compat.h -- lets callers written against the reference's headers (Eigen, Pangolin GL
// types) compile against the HIP core when those libraries are absent (this container, the
// GPU box).  With the real libraries present the real types are used unchanged.
#pragma once

#include <cstdint>
#include <cstring>
#include <utility>

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#else
namespace Eigen {
// minimal stand-in: column-major 4x4 float, the storage of Eigen::Matrix4f
struct Matrix4f {
    float m[16];
    Matrix4f() { std::memset(m, 0, sizeof m); }
    static Matrix4f Identity() { Matrix4f r; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
    float &operator()(int r, int c) { return m[c * 4 + r]; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    float *data() { return m; }
    const float *data() const { return m; }
};
}  // namespace Eigen
#endif

#if __has_include(<pangolin/gl/gl.h>)
#include <pangolin/gl/gl.h>
#else
typedef unsigned int GLuint;
#define SM_COMPAT_POD_TEXTURE 1
namespace pangolin {
// POD handle with the fields callers read (build_map.cpp:34-38 shows textures by pointer only) and, since no GL object
// exists here, the host copy of the image the reference's texture would hold (row-major; whichever matches the texture's type)
struct GlTexture {
    int width = 0, height = 0;
    GLuint tid = 0;
    const float *host = nullptr;               // DEPTH_METRIC / DEPTH_FILTERED / LAST: H x W metres
    const unsigned char *host_u8 = nullptr;    // RGB: H x W x 3; SEMANTIC: H x W
    const unsigned short *host_u16 = nullptr;  // DEPTH (raw): H x W millimetres
};
struct OpenGlMatrix { double m[16]; };
}  // namespace pangolin
#endif
