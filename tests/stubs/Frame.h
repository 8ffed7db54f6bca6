// TEST INFRASTRUCTURE -- stand-in for the reference header of the same name: see standin_orbslam3.hpp.
#pragma once
#include "standin_orbslam3.hpp"
