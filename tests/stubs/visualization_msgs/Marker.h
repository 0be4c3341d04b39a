#pragma once
#include <vector>
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
struct Vector3 { double x = 0, y = 0, z = 0; };
}
namespace std_msgs { struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; }
namespace visualization_msgs {
struct Marker {
    enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, ADD = 0, MODIFY = 0, DELETE = 2 };
    std_msgs::Header header;
    std::string ns; int32_t id = 0; int32_t type = 0; int32_t action = 0;
    geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color;
    std::vector<geometry_msgs::Point> points;
};
}
