#pragma once
#include "Index.h"
namespace faiss {
// src/faiss_extension.cpp:154-155
Index *index_factory(int d, const char *description, MetricType metric = METRIC_L2);
} // namespace faiss
