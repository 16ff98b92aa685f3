// scene_io.h -- flattened scenes on disk (scene_io.cpp; product code, host only).
#pragma once
#include <string>

#include "flatten.h"

// Every index a kernel follows stays inside its table, the BVH is acyclic and no deeper than recorded.
bool validate_flat_scene(const FlatScene &fs, std::string &err);
bool save_flat_scene(const FlatScene &fs, const char *path, std::string &err);
// Reads, checks the CRC and the table layouts, then validate_flat_scene(): a scene that loads is safe to upload.
bool load_flat_scene(const char *path, FlatScene &fs, std::string &err);
