// Instantiates the register-resident recursion kernel for 1 wave(s) per target.
#include "paint_regs.hpp"

namespace relate {
hipError_t launch_paint_regs_g1(const PaintArgs& a, int MPL, hipStream_t st) { return regs::launch_regs_g<1>(a, MPL, st); }
}  // namespace relate
