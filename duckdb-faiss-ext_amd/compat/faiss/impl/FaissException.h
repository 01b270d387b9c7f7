// compat/faiss/impl/FaissException.h -- caught BY VALUE by the glue (src/faiss_extension.cpp:397,514,584,632) and
// pattern-matched on .msg (:400,:523,:592; src/gpu/gpu.cpp:52,56).
#pragma once
#include <exception>
#include <string>
namespace faiss {
class FaissException : public std::exception {
public:
	explicit FaissException(const std::string &m) : msg(m) {
	}
	const char *what() const noexcept override {
		return msg.c_str();
	}
	std::string msg;
};
} // namespace faiss
