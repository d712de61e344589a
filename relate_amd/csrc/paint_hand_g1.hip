// Instantiates the hand-scheduled recursion kernel for 1 wave(s) per target.
#include "paint_hand.hpp"

namespace relate {
hipError_t launch_paint_hand_g1(const PaintArgs& a, int MPL, hipStream_t st) { return hand::launch_hand_g<1>(a, MPL, st); }
}  // namespace relate
