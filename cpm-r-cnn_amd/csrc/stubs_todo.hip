// temporary: symbols declared in the header and implemented in later files
#include "common.h"
CPM_EXPORT size_t cpm_conv2d_workspace_bytes(const cpm_conv_desc*) { return 0; }
CPM_EXPORT size_t cpm_stem_workspace_bytes(int, int, int) { return 0; }
