// Declarations of the part of Eigen's API that build_map.cpp / load_map.cpp / gui/GUI.h use (syntax check only).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
namespace Eigen {
template <class T> struct aligned_allocator : std::allocator<T> {
    aligned_allocator() = default;
    template <class U> aligned_allocator(const aligned_allocator<U> &) {}
    template <class U> struct rebind { typedef aligned_allocator<U> other; };
};
struct Vector2f { float v[2]; Vector2f() {} Vector2f(float, float) {} };
struct Vector3f {
    float v[3];
    Vector3f() { v[0] = v[1] = v[2] = 0; }
    Vector3f(float x, float y, float z) { v[0] = x; v[1] = y; v[2] = z; }
    float &operator()(int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    Vector3f normalized() const { return *this; }
    Vector3f cross(const Vector3f &) const { return *this; }
    float dot(const Vector3f &) const { return 0.f; }
    float norm() const { return 0.f; }
    Vector3f operator-(const Vector3f &) const { return *this; }
    Vector3f operator+(const Vector3f &) const { return *this; }
    Vector3f operator-() const { return *this; }
    float *data() { return v; }
};
struct Vector4f { float v[4]; Vector4f() {} Vector4f(float, float, float, float) {} };
template <class S> struct AngleAxis { AngleAxis(S, const Vector3f &) {} };
struct Matrix3f {
    float m[9];
    Matrix3f() {}
    template <class S> Matrix3f(const AngleAxis<S> &) {}
    template <class S> Matrix3f &operator=(const AngleAxis<S> &) { return *this; }
    Vector3f operator*(const Vector3f &v) const { return v; }
};
struct Matrix4f {
    float m[16];
    Matrix4f() { std::memset(m, 0, sizeof m); }
    static Matrix4f Identity() { Matrix4f r; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
    float &operator()(int r, int c) { return m[c * 4 + r]; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    float *data() { return m; }
    const float *data() const { return m; }
    Matrix3f topLeftCorner(int, int) const { return Matrix3f(); }
    template <int R, int C> Vector3f topRightCorner() const { return Vector3f(); }
    Matrix4f operator*(const Matrix4f &) const { return *this; }
    Matrix4f inverse() const { return *this; }
};
struct Matrix4d {
    double m[16];
    struct Comma { Comma &operator,(double) { return *this; } };
    Comma operator<<(double) { return Comma(); }
    double *data() { return m; }
};
enum { Affine = 1 };
template <class S, int D, int M> struct Transform {
    Matrix4f matrix() const { return Matrix4f(); }
};
struct Translation3f {
    Translation3f(float, float, float) {}
    template <class S> Transform<float, 3, Affine> operator*(const AngleAxis<S> &) const { return Transform<float, 3, Affine>(); }
};
}  // namespace Eigen
