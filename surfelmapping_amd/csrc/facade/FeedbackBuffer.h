// FeedbackBuffer.h -- drop-in for src/FeedbackBuffer.h:32-65 over the C-ABI: the raw per-frame surfel cloud
// (surfel_feedback.vert) that build_map.cpp:177-184 draws through getFeedbackBuffer(FeedbackBuffer::RAW)->render(...).
// The cloud is computed by the HIP core on demand (sm_download_raw_cloud) and kept on the host as the reference's
// interleaved 12-float vertices; built with SM_FACADE_GL (Pangolin / GL present) render() also draws it, as GL_POINTS
// from a vertex buffer in the reference's layout (position at offset 0, stride Config::vertexSize()).
#pragma once
#include <string>
#include <vector>
#include "../../../include/sm_c_api.h"
#include "Config.h"
#include "sm_compat.h"

class FeedbackBuffer {
public:
    explicit FeedbackBuffer(sm_ctx *ctx = nullptr) : vbo(0), fid(0), ctx_(ctx) {}
    void bind(sm_ctx *ctx) { ctx_ = ctx; }

    // src/FeedbackBuffer.cpp:85-145 computes the cloud on the GPU every frame; here refresh() pulls it when somebody looks
    unsigned int refresh()
    {
        uint32_t n = 0;
        if (!ctx_ || sm_download_raw_cloud(ctx_, nullptr, 0, &n) != SM_OK) { vertices_.clear(); return 0; }
        vertices_.resize((size_t)n * 12);
        if (n && sm_download_raw_cloud(ctx_, vertices_.data(), n, &n) != SM_OK) vertices_.clear();
        return (unsigned int)(vertices_.size() / 12);
    }

    // src/FeedbackBuffer.cpp:147-200.  `pose` moves the camera-frame cloud to the world frame in the reference's shader;
    // the flags select its colouring (normals / colours / classes / discs).
    void render(pangolin::OpenGlMatrix mvp, const Eigen::Matrix4f &pose, const bool drawNormals, const bool drawColors,
                const bool drawSemantic, const bool drawSurfel)
    {
        (void)mvp; (void)pose; (void)drawNormals; (void)drawColors; (void)drawSemantic; (void)drawSurfel;
        const unsigned int n = refresh();
#ifdef SM_FACADE_GL
        if (!n) return;
        if (!vbo) glGenBuffers(1, &vbo);
        glBindBuffer(GL_ARRAY_BUFFER, vbo);
        glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)vertices_.size() * sizeof(float), vertices_.data(), GL_STREAM_DRAW);
        glMatrixMode(GL_PROJECTION); glLoadIdentity(); glMultMatrixd(mvp.m);
        glMatrixMode(GL_MODELVIEW); glLoadIdentity(); glMultMatrixf(pose.data());
        glEnableClientState(GL_VERTEX_ARRAY);
        glVertexPointer(3, GL_FLOAT, Config::vertexSize(), 0);
        glDrawArrays(GL_POINTS, 0, (GLsizei)n);
        glDisableClientState(GL_VERTEX_ARRAY);
        glBindBuffer(GL_ARRAY_BUFFER, 0);
#else
        (void)n;
#endif
    }

    static const std::string RAW, FILTERED;
    GLuint vbo;
    GLuint fid;

    // host copy: count() surfels of 12 floats (pos, 0.9 | colour bits, 0, time, time | normal, radius), camera frame
    const std::vector<float> &vertices() const { return vertices_; }
    unsigned int count() const { return (unsigned int)(vertices_.size() / 12); }

private:
    sm_ctx *ctx_;
    std::vector<float> vertices_;
};
inline const std::string FeedbackBuffer::RAW = "RAW";              // src/FeedbackBuffer.cpp:21-22
inline const std::string FeedbackBuffer::FILTERED = "FILTERED";
