#include "../mock_ros_core.h"
