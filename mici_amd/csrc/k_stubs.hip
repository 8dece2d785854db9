// Temporary launch stubs (replaced as kernels land).
#include "mm_internal.h"
