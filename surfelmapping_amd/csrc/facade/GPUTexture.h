// GPUTexture.h -- name-compatible handle (src/GPUTexture.h): the device images live inside the
// HIP context; `texture` is a POD the GUI can fill/read lazily.
#pragma once
#include <string>
#include <vector>
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
    // host copies of what the reference keeps in the GL texture (src/SurfelMapping.cpp:50-85: RGB u8 x 3, DEPTH u16, SEMANTIC u8,
    // DEPTH_FILTERED / DEPTH_METRIC / LAST f32); SurfelMapping::getTexture() fills them and, with SM_FACADE_GL, uploads them
    std::vector<float> host_f;
    std::vector<unsigned char> host_u8;
    std::vector<unsigned short> host_u16;
private:
    pangolin::GlTexture tex_;
};
