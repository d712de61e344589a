// Instantiations of the painting kernel with the assembly steady-state loop for 2 wave(s) per target.
#include "paint_loop.hpp"

namespace relate {

hipError_t launch_paint_loop_g2(const PaintArgs& a, int MPL, hipStream_t st) { return aloop::launch_loop_g<2>(a, MPL, st); }

}  // namespace relate
