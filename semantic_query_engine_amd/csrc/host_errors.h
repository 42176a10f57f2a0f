// host_errors.h -- error plumbing of the C ABI (thread-local last-error text, api.hip).  Plain C++: the host-only
// tokenizer includes this and not common.h, so that it also builds with g++ under the sanitizers (make tokenizer_asan).
#pragma once

#include <string>

#include "sqe.h"

namespace sqe {
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
}  // namespace sqe
