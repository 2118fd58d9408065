// identity of the sources this library was built from (see Makefile: BUILD_ID)
#include "build_id.h"
extern "C" const char* pyvb_build_id(void) { return PYVB_BUILD_ID; }
