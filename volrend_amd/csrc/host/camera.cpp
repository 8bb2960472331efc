// camera.cpp -- volrend::Camera (pose model of the reference's src/camera.cpp:26-58).
#include "volrend/camera.hpp"

namespace volrend {

Camera::Camera(int width, int height, float fx, float fy)
    : width(width),
      height(height),
      fx(fx < 0.f ? CAMERA_DEFAULT_FOCAL_LENGTH : fx),
      fy(fy < 0.f ? this->fx : fy) {
    center = glm::vec3(-3.55f, 0.0f, 3.55f);
    v_back = glm::vec3(-0.7071068f, 0.0f, 0.7071068f);
    v_world_up = glm::vec3(0.0f, 0.0f, 1.0f);
    origin = glm::vec3(0.0f, 0.0f, 0.0f);
    _update();
}

void Camera::_update(bool transform_from_vecs, bool /*copy_device*/) {
    if (transform_from_vecs) {
        v_back = glm::normalize(v_back);
        v_right = glm::normalize(glm::cross(v_world_up, v_back));
        v_up = glm::cross(v_back, v_right);
        transform[0] = v_right;
        transform[1] = v_up;
        transform[2] = v_back;
        transform[3] = center;
    }
    // The projection / w2c matrices of the reference feed the GL mesh rasteriser only.
}

}  // namespace volrend
