#pragma once
#include <vector>
#include <sensor_msgs/PointField.h>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct PointCloud2 {
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::vector<PointField> fields;
    bool is_bigendian = false;
    uint32_t point_step = 0, row_step = 0;
    std::vector<uint8_t> data;
    bool is_dense = false;
};
typedef gm_stub::shared_ptr<PointCloud2 const> PointCloud2ConstPtr;
}
