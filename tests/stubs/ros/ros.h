// Stand-in for <ros/ros.h>: only what ros/geometric_mapping_node.cpp uses.  Syntax check only (tests/stubs/README.md).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <boost_stub_shared_ptr.h>
namespace ros {
struct Time { uint32_t sec = 0, nsec = 0; static Time now() { return Time(); } };
class Publisher {
public:
    template <class M> void publish(const M &) const {}
};
class Subscriber {};
struct Duration { explicit Duration(double) {} };
struct TimerEvent {};
class Timer {};
class NodeHandle {
public:
    bool getParam(const std::string &, double &) const { return false; }
    bool getParam(const std::string &, bool &) const { return false; }
    bool getParam(const std::string &, std::string &) const { return false; }
    template <class M> Publisher advertise(const std::string &, uint32_t) { return Publisher(); }
    Timer createTimer(Duration, void (*)(const TimerEvent &)) { return Timer(); }
    template <class M> Subscriber subscribe(const std::string &, uint32_t, void (*)(const gm_stub::shared_ptr<M const> &)) { return Subscriber(); }
};
inline void init(int &, char **, const std::string &) {}
inline void spin() {}
}  // namespace ros
#define ROS_INFO(...) std::printf(__VA_ARGS__)
#define ROS_WARN(...) std::printf(__VA_ARGS__)
#define ROS_ERROR(...) std::printf(__VA_ARGS__)
#define ROS_FATAL(...) std::printf(__VA_ARGS__)
