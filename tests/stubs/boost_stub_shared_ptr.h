// Stand-in for boost::shared_ptr as roscpp's ConstPtr typedefs use it.  Syntax check only.
#pragma once
#include <memory>
namespace gm_stub { template <class T> using shared_ptr = std::shared_ptr<T>; }
