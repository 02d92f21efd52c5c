#include "mpeg.hpp"
namespace mpeg {
std::unique_ptr<VideoBackend> Device::newVideoBackend() { return nullptr; }
std::unique_ptr<AudioBackend> Device::newAudioBackend(int) { return nullptr; }
std::unique_ptr<BatchStore> Device::newBatchStore() { return nullptr; }
std::unique_ptr<AudioBatchStore> Device::newAudioBatchStore() { return nullptr; }
int Device::NumaNode() const { return -1; }
}
