// Instantiations of the painting kernel with the assembly steady-state loop for 8 wave(s) per target.
#include "paint_loop.hpp"

namespace relate {

hipError_t launch_paint_loop_g8(const PaintArgs& a, int MPL, hipStream_t st) { return aloop::launch_loop_g<8>(a, MPL, st); }

}  // namespace relate
