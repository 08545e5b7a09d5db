#define QOI_IMPLEMENTATION
#include "qoi.h"
