// instantiation unit: two object slots and four grid cells per lane (D <= 128, W*H <= 256), e.g. the 16x16 level
#include "cz_kernels.h"
namespace cz {
Launchers launchers_large() { return Launchers{&Inst<2, 4>::step, &Inst<2, 4>::reset, &Inst<2, 4>::observe}; }
}
