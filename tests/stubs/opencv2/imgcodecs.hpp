// OpenCV is only #included by build_map.cpp / load_map.cpp, never used by name there (syntax check only)
