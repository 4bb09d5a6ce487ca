// instantiation unit: one object slot and one grid cell per lane (D <= 64, W*H <= 64), e.g. the 7x7 levels
#include "cz_kernels.h"
namespace cz {
Launchers launchers_small() { return Launchers{&Inst<1, 1>::step, &Inst<1, 1>::reset, &Inst<1, 1>::observe}; }
}
