// volrend/vecmath.hpp -- the handful of glm types the kept volrend headers use
// (reference include/volrend/camera.hpp:5-8, n3tree.hpp:13).  The reference's glm is an
// un-vendored submodule; define VOLREND_USE_GLM to use a real glm instead.
#pragma once
#ifdef VOLREND_USE_GLM
#include <glm/mat4x3.hpp>
#include <glm/mat4x4.hpp>
#include <glm/vec2.hpp>
#include <glm/vec3.hpp>
#else
#include <cmath>
namespace glm {
struct vec2 {
    float x = 0, y = 0;
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(const vec3& a, const vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(const vec3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(const vec3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3& a, const vec3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline vec3 normalize(const vec3& a) { return a / std::sqrt(dot(a, a)); }
// column-major 4 columns x 3 rows, m[col][row] like glm::mat4x3
struct mat4x3 {
    vec3 c[4];
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
inline const float* value_ptr(const mat4x3& m) { return &m.c[0].x; }
}  // namespace glm
#endif
