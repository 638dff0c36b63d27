// j40_amd/csrc/capi.hpp -- the object behind the opaque j40hip_frame handle
#pragma once
#include "../../include/j40hip.h"
#include "frame.hpp"

struct j40hip_device_state;  // defined in device/runtime.hip

struct j40hip_frame {
	const uint8_t *cs = nullptr;     // codestream bytes (inside the caller's buffer, or cs_storage)
	size_t cs_size = 0;
	std::vector<uint8_t> cs_storage;
	j40hip::Frame frame;
	j40hip_device_state *dev = nullptr;
};

// implemented next to the kernels; a no-op when nothing was uploaded
extern "C" void j40hip_release_device(j40hip_frame *f);
