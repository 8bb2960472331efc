// volrend::Camera -- pose + intrinsics of the reference's include/volrend/camera.hpp
// (the GUI drag helpers are out of scope for the headless path).  The 48-byte device
// copy of `transform` (reference src/camera.cpp:67-75) does not exist any more: the pose
// rides in the launch arguments.
#pragma once
#include "volrend/common.hpp"
#include "volrend/vecmath.hpp"

namespace volrend {
static const float CAMERA_DEFAULT_FOCAL_LENGTH = 1111.11f;

struct Camera {
    Camera(int width = 256, int height = 256, float fx = CAMERA_DEFAULT_FOCAL_LENGTH,
           float fy = -1.f);

    // Camera pose model, you can modify these
    glm::vec3 v_back, v_world_up, center;
    // Origin for about-origin rotation
    glm::vec3 origin;
    // Vectors below are automatically updated
    glm::vec3 v_up, v_right;
    // 4x3 C2W transform used for volume rendering (columns right, up, back, centre)
    glm::mat4x3 transform;

    // Image size
    int width, height;
    // Focal length
    float fx, fy;

    // Update the transform after modifying v_right/v_forward/center (reference
    // src/camera.cpp:47-58).  `copy_device` is accepted for source compatibility and ignored.
    void _update(bool transform_from_vecs = true, bool copy_device = true);
};

}  // namespace volrend
