// GPUTexture.h -- name-compatible handle (src/GPUTexture.h): the device images live inside the
// HIP context; `texture` is a POD the GUI can fill/read lazily.
#pragma once
#include <string>
#include "sm_compat.h"

class GPUTexture {
public:
    GPUTexture() : texture(&tex_) {}
    pangolin::GlTexture *texture;
    static constexpr const char *RGB = "RGB";
    static constexpr const char *DEPTH_RAW = "DEPTH";           // sic: src/GPUTexture.cpp:22
    static constexpr const char *DEPTH_FILTERED = "DEPTH_FILTERED";
    static constexpr const char *DEPTH_METRIC = "DEPTH_METRIC";
    static constexpr const char *SEMANTIC = "SEMANTIC";
private:
    pangolin::GlTexture tex_;
};
