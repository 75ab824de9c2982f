// PathArgs for host translation units that must not see the device code of path_core.h.
#ifndef BLOK_PATH_ARGS_H
#define BLOK_PATH_ARGS_H
#include "path_core.h"
#endif
