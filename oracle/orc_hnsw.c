/* oracle/orc_hnsw.c -- placeholder translation unit; the HNSW restatement lands here. */
#include "orc.h"
