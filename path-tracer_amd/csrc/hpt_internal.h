// hpt_internal.h -- seams between hpt_api.cpp (contexts, scenes, the one-GPU render driver) and hpt_group.cpp (several
// GPUs in one process).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hpt.h"

// copies the host-side description of a committed scene (meshes, trees, nodes, materials, images, camera) onto another
// context and commits it there: the scene is REPLICATED on every GPU of a group
int hpti_scene_clone(const hpt_scene* src, hpt_context* ctx, hpt_scene** out);
uint64_t hpti_scene_generation(const hpt_scene* s);      // counts hpt_scene_commit calls
hpt_context* hpti_scene_context(const hpt_scene* s);
void hpti_scene_frame(const hpt_scene* s, uint32_t* w, uint32_t* h);
hipStream_t hpti_context_stream(const hpt_context* c);
int hpti_context_device(const hpt_context* c);
int hpti_context_max_blocks(const hpt_context* c);
int hpti_fail(int code, const char* fmt, ...);           // sets the calling thread's hpt_last_error() text, returns code
