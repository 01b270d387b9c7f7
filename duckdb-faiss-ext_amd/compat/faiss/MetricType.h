// compat/faiss/MetricType.h -- faiss::MetricType / idx_t as used by /root/reference/src/faiss_extension.cpp:54-68.
#pragma once
#include <cstdint>
namespace faiss {
using idx_t = int64_t;
enum MetricType {
	METRIC_INNER_PRODUCT = 0,
	METRIC_L2 = 1,
	METRIC_L1,
	METRIC_Linf,
	METRIC_Lp,
	METRIC_Canberra = 20,
	METRIC_BrayCurtis,
	METRIC_JensenShannon,
	METRIC_Jaccard,
};
} // namespace faiss
