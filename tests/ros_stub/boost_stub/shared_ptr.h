// tests/ros_stub: roscpp message pointers are boost::shared_ptr<M const>; for a syntax / link check std::shared_ptr has the same surface.
#pragma once
#include <memory>
namespace boost
{
template <class T> using shared_ptr = std::shared_ptr<T>;
}
