// TEST INFRASTRUCTURE: forwards to the oracle/_ref host shim (see prcnn_ref_shim.h)
#include "prcnn_ref_shim.h"
