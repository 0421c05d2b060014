// gui/GUI.h includes <Shaders.h> for the Shader handle type and, through it, Eigen (syntax check only).
#pragma once
#include <memory>
#include <Eigen/Core>
class Shader {};
