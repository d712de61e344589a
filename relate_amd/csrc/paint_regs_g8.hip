// Instantiates the register-resident recursion and re-painting kernels for 8 wave(s) per target.
// Several waves per target: the compiler keeps to s0..s43 so that a round of row words holds 20 of them (one round
// and one wait per site at 20 words per lane, two at 40) -- measured on the 5008-haplotype chunk.
#define RELATE_CS 44
#include "repaint_regs.hpp"

namespace relate {
hipError_t launch_paint_regs_g8(const PaintArgs& a, int MPL, hipStream_t st) { return regs::launch_regs_g<8>(a, MPL, st); }
hipError_t launch_repaint_regs_g8(const RepaintLaunch& a, int MPL, hipStream_t st)
{
  return regs::launch_repaint_g<8>(a, MPL, st);
}
}  // namespace relate
