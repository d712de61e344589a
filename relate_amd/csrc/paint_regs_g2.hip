// Instantiates the register-resident recursion and re-painting kernels for 2 wave(s) per target.
#include "repaint_regs.hpp"

namespace relate {
hipError_t launch_paint_regs_g2(const PaintArgs& a, int MPL, hipStream_t st) { return regs::launch_regs_g<2>(a, MPL, st); }
hipError_t launch_repaint_regs_g2(const RepaintLaunch& a, int MPL, hipStream_t st)
{
  return regs::launch_repaint_g<2>(a, MPL, st);
}
}  // namespace relate
