// Layout constants shared by the normal-equation kernels (ecal_solver.hip) and the host half of the LM step (arrow_host.hpp,
// which also compiles without HIP: the ThreadSanitizer build of tests/cpp/tsan_host_half.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

namespace ecal {

// accumulation buffer: [0] cost | [1..9] g_intr | [10..90] H_intr (9x9, upper) | per control point c at
// 91 + 204 c: g_c[6] | H_c,intr[6][9] | H_c,c+d[4][6][6] (d = 0..3; d = 0 upper only)
constexpr size_t ACC_HEAD = 91, ACC_PER_CP = 204;

// streamed evaluation (NeProgress): at most this many groups of chunks = interiors of the host's partition
constexpr int NE_MAX_GROUPS = 32;

}  // namespace ecal
