#include "../../include/mireg.h"
extern "C" {
int mireg_version(void) { return 2; }
const char* mireg_arch(void) { return "gfx950"; }
}
