// kernels_tiled.hpp -- LDS-tiled fused level kernels (production path).
#pragma once
#include "common.hpp"
