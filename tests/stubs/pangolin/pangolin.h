#include "../pangolin_stub.h"
