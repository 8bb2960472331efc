/* Minimal stand-in for the (un-vendored, empty-submodule) glm headers: just
 * enough type shape for the reference's camera.hpp / n3tree.hpp member
 * declarations.  ORACLE / test infrastructure only. */
#ifndef VR_REF_SHIM_GLM_H_
#define VR_REF_SHIM_GLM_H_
namespace glm {
struct vec2 {
    float x = 0, y = 0;
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
};
struct mat4x3 {
    vec3 c[4];
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
struct mat4x4 {
    vec4 c[4];
};
typedef mat4x4 mat4;
}  // namespace glm
#endif
