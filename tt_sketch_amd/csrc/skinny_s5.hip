#include "skinny_inst.h"
namespace ttsk {
template int launch_skinny_s_depth<5, 0>(const SkinnyS &, int, int, size_t, int, hipStream_t);
}
