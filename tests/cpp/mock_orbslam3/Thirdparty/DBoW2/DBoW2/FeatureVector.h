// TEST INFRASTRUCTURE (see ../../../README.md): DBoW2::FeatureVector is a std::map<NodeId, std::vector<unsigned int>> (FeatureVector.h:24-29)
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
}  // namespace DBoW2
