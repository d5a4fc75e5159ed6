// TEST INFRASTRUCTURE ONLY: see lslam_ros_pcl_shim.hpp
#include "lslam_ros_pcl_shim.hpp"
