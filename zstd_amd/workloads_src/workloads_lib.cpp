// workloads_lib.cpp — bench / test INPUT GENERATORS (not product): built into zstd_amd/libzhip_workloads.so by zstd_amd/build.py.
// libzstd_hip.so exports nothing from here.
#include "zhip_datagen.h"

extern "C" void zhip_wl_datagen(void* buffer, size_t size, double matchProba, double litProba, unsigned seed, int streamMode)
{
    zhip::datagen(buffer, size, matchProba, litProba, seed, streamMode);
}
