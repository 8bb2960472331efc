#include "glm/_shim.hpp"
