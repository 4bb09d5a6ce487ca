// instantiation unit: four object slots and sixteen grid cells per lane (D <= 255, W*H <= 1024), the reference's whole
// level range.  Same code as the other two units; at this size the per-lane state no longer fits the scalar register file
// (the compiler spills a few hundred SGPRs), so a step costs several times the 7x7 one -- capacity, not speed.
#include "cz_kernels.h"
namespace cz {
Launchers launchers_huge() { return Launchers{&Inst<4, 16>::step, &Inst<4, 16>::reset, &Inst<4, 16>::observe}; }
}
