// TEST INFRASTRUCTURE -- stand-in for the fork's IMU sample type (include/Socket/imudata.h:10-20: a time stamp and two float triples).
#pragma once
#include <vector>
class IMUData {
public:
    long ts_;
    std::vector<float> gyro_, acce_;
    IMUData(long ts, std::vector<float>& gyro, std::vector<float>& acce) : ts_(ts), gyro_(gyro), acce_(acce) {}
};
