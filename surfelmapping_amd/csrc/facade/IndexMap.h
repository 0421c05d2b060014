// IndexMap.h -- drop-in for src/IndexMap.h:28-88 over the C-ABI.
#pragma once
#include <vector>
#include "../../../include/sm_c_api.h"
#include "GPUTexture.h"
#include "sm_compat.h"

class IndexMap {
public:
    static const int FACTOR = 1;                     // src/IndexMap.cpp:21
    explicit IndexMap(sm_ctx *ctx = nullptr) : ctx_(ctx) {}
    void bind(sm_ctx *ctx) { ctx_ = ctx; }

    // src/IndexMap.cpp:138-198; `model` (vbo id, count) is implicit: the context owns the model
    void predictIndices(const Eigen::Matrix4f &pose, const int &time, const std::pair<GLuint, GLuint> & /*model*/,
                        const float depthCutoff, const int timeDelta)
    {
        sm_stage_splat(ctx_, pose.data(), time, depthCutoff, timeDelta);
    }

    GPUTexture *indexTex() { return &indexTexture; }
    GPUTexture *vertConfTex() { return &vertConfTexture; }
    GPUTexture *colorTimeTex() { return &colorTimeTexture; }
    GPUTexture *normalRadTex() { return &normalRadTexture; }

    // read-back of the four index-map images (row-major H*W)
    int download(std::vector<int32_t> &id, std::vector<float> &vc, std::vector<float> &ct, std::vector<float> &nr, int P)
    {
        id.resize(P); vc.resize((size_t)P * 4); ct.resize((size_t)P * 4); nr.resize((size_t)P * 4);
        return sm_download_index_map(ctx_, id.data(), vc.data(), ct.data(), nr.data());
    }

private:
    sm_ctx *ctx_;
    GPUTexture indexTexture, vertConfTexture, colorTimeTexture, normalRadTexture;
};
