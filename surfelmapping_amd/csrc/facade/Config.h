// Config.h -- drop-in for the reference's process-wide Config singleton (src/Config.h:15-58,
// src/Config.cpp:7-38): same static accessors, same defaults.
#pragma once
#include <string>

class Config {
public:
    static Config &getInstance(float fx = 0, float fy = 0, float cx = 0, float cy = 0, int rows = 0, int cols = 0)
    {
        static Config instance(fx, fy, cx, cy, rows, cols);   // first call fixes the values (src/Config.cpp:40-44)
        return instance;
    }
    static float &fx() { return getInstance().fx_; }
    static float &fy() { return getInstance().fy_; }
    static float &cx() { return getInstance().cx_; }
    static float &cy() { return getInstance().cy_; }
    static int &H() { return getInstance().rows_; }
    static int &W() { return getInstance().cols_; }
    static int &numPixels() { return getInstance().num_pixels; }
    static int &vertexSize() { return getInstance().vertex_size; }
    static float &nearClip() { return getInstance().near_clip; }
    static float &farClip() { return getInstance().far_clip; }
    static float &surfelFuseDistanceThreshFactor() { return getInstance().surfel_fuse_distance_threshold_factor; }
    static int &maxSqrtVertices() { return getInstance().max_sqrt_vertices; }
    static std::string shaderDir() { return ""; }   // no runtime-compiled shaders in this core

private:
    float fx_, fy_, cx_, cy_;
    int rows_, cols_, num_pixels, vertex_size;
    float near_clip, far_clip, surfel_fuse_distance_threshold_factor;
    int max_sqrt_vertices;
    Config(float fx, float fy, float cx, float cy, int rows, int cols)
        : fx_(fx), fy_(fy), cx_(cx), cy_(cy), rows_(rows), cols_(cols), num_pixels(rows * cols),
          vertex_size(48), near_clip(1.0f), far_clip(30.0f), surfel_fuse_distance_threshold_factor(0.0f),
          max_sqrt_vertices(5000) {}
};
