// dmap_io.cpp -- see include/dmapio.h.  Plain C++ (no GPU): the .dmap file is the reference's
// checkpoint / inter-process exchange format (SceneDensify.cpp:2010,2095-2117), kept byte-identical.
#include "../../include/dmapio.h"
#include <stdio.h>
#include <string.h>
#include <string>

namespace {
#pragma pack(push, 1)
struct RawHeader { uint16_t name; uint8_t type; uint8_t padding; uint32_t imageWidth, imageHeight, depthWidth, depthHeight; float dMin, dMax; };
#pragma pack(pop)
static_assert(sizeof(RawHeader) == 28, "HeaderDepthDataRaw is 28 bytes");
const uint16_t kName = (uint16_t)('D' | ('R' << 8));
}

extern "C" {

int dmap_write(const char* fileName, const DMapHeader* h, const float* depth, const float* normal, const float* conf, const uint8_t* views) {
	if (!fileName || !h || !depth || h->nIDs < 2 || h->nIDs > 255 || h->depthWidth == 0 || h->depthHeight == 0 ||
		h->depthWidth > h->imageWidth || h->depthHeight > h->imageHeight) return -2;
	const std::string tmp = std::string(fileName) + ".tmp";       // DepthData::Save, DepthMap.cpp:234-252
	FILE* f = fopen(tmp.c_str(), "wb");
	if (!f) return -1;
	RawHeader rh; memset(&rh, 0, sizeof(rh));
	rh.name = kName;
	rh.type = (uint8_t)(1u | (normal ? 2u : 0u) | (conf ? 4u : 0u) | (views ? 8u : 0u));
	rh.imageWidth = h->imageWidth; rh.imageHeight = h->imageHeight; rh.depthWidth = h->depthWidth; rh.depthHeight = h->depthHeight;
	rh.dMin = h->dMin; rh.dMax = h->dMax;
	const size_t n = (size_t)h->depthWidth * h->depthHeight;
	const uint16_t nameLen = (uint16_t)strnlen(h->imageFileName, sizeof(h->imageFileName));
	bool ok = fwrite(&rh, sizeof(rh), 1, f) == 1;
	ok = ok && fwrite(&nameLen, 2, 1, f) == 1 && (nameLen == 0 || fwrite(h->imageFileName, 1, nameLen, f) == nameLen);
	ok = ok && fwrite(&h->nIDs, 4, 1, f) == 1 && fwrite(h->IDs, 4, h->nIDs, f) == h->nIDs;
	ok = ok && fwrite(h->K, 8, 9, f) == 9 && fwrite(h->R, 8, 9, f) == 9 && fwrite(h->C, 8, 3, f) == 3;
	ok = ok && fwrite(depth, 4, n, f) == n;
	if (normal) ok = ok && fwrite(normal, 12, n, f) == n;
	if (conf) ok = ok && fwrite(conf, 4, n, f) == n;
	if (views) ok = ok && fwrite(views, 4, n, f) == n;
	ok = (fclose(f) == 0) && ok;
	if (!ok || rename(tmp.c_str(), fileName) != 0) { remove(tmp.c_str()); return -1; }
	return 0;
}

static int readHeader(FILE* f, DMapHeader* h) {
	RawHeader rh;
	if (fread(&rh, sizeof(rh), 1, f) != 1 || rh.name != kName || (rh.type & 1u) == 0 || rh.depthWidth == 0 || rh.depthHeight == 0 ||
		rh.imageWidth < rh.depthWidth || rh.imageHeight < rh.depthHeight) return -2;   // ImportDepthDataRaw validity test, DepthMap.cpp:1962-1970
	memset(h, 0, sizeof(*h));
	h->imageWidth = rh.imageWidth; h->imageHeight = rh.imageHeight; h->depthWidth = rh.depthWidth; h->depthHeight = rh.depthHeight;
	h->dMin = rh.dMin; h->dMax = rh.dMax; h->type = rh.type;
	uint16_t nameLen = 0;
	if (fread(&nameLen, 2, 1, f) != 1) return -2;
	std::string name(nameLen, '\0');
	if (nameLen && fread(&name[0], 1, nameLen, f) != nameLen) return -2;
	strncpy(h->imageFileName, name.c_str(), sizeof(h->imageFileName) - 1);
	if (fread(&h->nIDs, 4, 1, f) != 1 || h->nIDs == 0 || h->nIDs > 255) return -2;
	if (fread(h->IDs, 4, h->nIDs, f) != h->nIDs) return -2;
	if (fread(h->K, 8, 9, f) != 9 || fread(h->R, 8, 9, f) != 9 || fread(h->C, 8, 3, f) != 3) return -2;
	return 0;
}

int dmap_read_header(const char* fileName, DMapHeader* h) {
	if (!fileName || !h) return -2;
	FILE* f = fopen(fileName, "rb");
	if (!f) return -1;
	const int rc = readHeader(f, h);
	fclose(f);
	return rc;
}

int dmap_read(const char* fileName, DMapHeader* h, unsigned flags, float* depth, float* normal, float* conf, uint8_t* views) {
	if (!fileName || !h) return -2;
	FILE* f = fopen(fileName, "rb");
	if (!f) return -1;
	int rc = readHeader(f, h);
	if (rc) { fclose(f); return rc; }
	const size_t n = (size_t)h->depthWidth * h->depthHeight;
	bool ok = true;
	auto plane = [&](unsigned bit, void* dst, size_t elem) {
		if (!(h->type & bit)) return;
		if ((flags & bit) && dst) ok = ok && fread(dst, elem, n, f) == n;
		else ok = ok && fseek(f, (long)(elem * n), SEEK_CUR) == 0;
	};
	plane(1, depth, 4); plane(2, normal, 12); plane(4, conf, 4); plane(8, views, 4);
	fclose(f);
	return ok ? 0 : -2;
}

} // extern "C"

// ---- .dimap: the disparity-data file of the SGM path (SemiGlobalMatcher::ExportDisparityDataRawFull / ImportDisparityDataRawFull,
// libs/MVS/SemiGlobalMatcher.cpp:2094-2188).  Layout: i32 imageW, imageH | f64 H[9] | f64 Q[16] | i16 subpixelSteps | i32 cols, rows |
// i16 disparity[rows*cols] | optional u16 cost[rows*cols]; the stored maps carry the 3-pixel border (NO_DISP / NO_ACCUMCOST) around the valid grid.
extern "C" {

int dimap_write(const char* fileName, int imageW, int imageH, const double H[9], const double Q[16], int16_t subpixelSteps,
		const int16_t* disparity, const uint16_t* cost, int w, int h) {
	if (!fileName || !H || !Q || !disparity || w <= 0 || h <= 0) return -2;
	const int HWB = 3, fw = w + 2 * HWB, fh = h + 2 * HWB;
	std::string tmp = std::string(fileName) + ".tmp";
	FILE* f = fopen(tmp.c_str(), "wb");
	if (!f) return -1;
	bool ok = fwrite(&imageW, 4, 1, f) == 1 && fwrite(&imageH, 4, 1, f) == 1 && fwrite(H, 8, 9, f) == 9 && fwrite(Q, 8, 16, f) == 16 &&
	          fwrite(&subpixelSteps, 2, 1, f) == 1 && fwrite(&fw, 4, 1, f) == 1 && fwrite(&fh, 4, 1, f) == 1;
	std::string row((size_t)fw * 2, 0);
	for (int pass = 0; pass < (cost ? 2 : 1) && ok; ++pass) {
		const uint16_t fill = pass == 0 ? (uint16_t)32767 : (uint16_t)65535;            // NO_DISP / NO_ACCUMCOST
		for (int y = 0; y < fh && ok; ++y) {
			uint16_t* r = (uint16_t*)&row[0];
			for (int x = 0; x < fw; ++x) r[x] = fill;
			const int sy = y - HWB;
			if (sy >= 0 && sy < h) memcpy(r + HWB, pass == 0 ? (const void*)(disparity + (size_t)sy * w) : (const void*)(cost + (size_t)sy * w), (size_t)w * 2);
			ok = fwrite(r, 2, (size_t)fw, f) == (size_t)fw;
		}
	}
	ok = (fclose(f) == 0) && ok;
	if (!ok || rename(tmp.c_str(), fileName) != 0) { remove(tmp.c_str()); return -1; }
	return 0;
}

// First call with disparity == NULL to get the sizes (w, h = valid grid; *hasCost); then with buffers of w*h entries.
int dimap_read(const char* fileName, int* imageW, int* imageH, double H[9], double Q[16], int16_t* subpixelSteps, int* w, int* h, int* hasCost,
		int16_t* disparity, uint16_t* cost) {
	if (!fileName) return -2;
	FILE* f = fopen(fileName, "rb");
	if (!f) return -1;
	int iw = 0, ih = 0, fw = 0, fh = 0; double hh[9], qq[16]; int16_t st = 0;
	bool ok = fread(&iw, 4, 1, f) == 1 && fread(&ih, 4, 1, f) == 1 && fread(hh, 8, 9, f) == 9 && fread(qq, 8, 16, f) == 16 && fread(&st, 2, 1, f) == 1 &&
	          fread(&fw, 4, 1, f) == 1 && fread(&fh, 4, 1, f) == 1;
	const int HWB = 3;
	if (!ok || iw <= 0 || ih <= 0 || fw <= 2 * HWB || fh <= 2 * HWB) { fclose(f); return -2; }
	const long dataStart = ftell(f);
	fseek(f, 0, SEEK_END); const long size = ftell(f);
	const long need = (long)fw * fh * 2;
	if (size - dataStart < need) { fclose(f); return -2; }
	const bool withCost = size - dataStart >= 2 * need;
	if (imageW) *imageW = iw; if (imageH) *imageH = ih; if (H) memcpy(H, hh, 72); if (Q) memcpy(Q, qq, 128); if (subpixelSteps) *subpixelSteps = st;
	if (w) *w = fw - 2 * HWB; if (h) *h = fh - 2 * HWB; if (hasCost) *hasCost = withCost ? 1 : 0;
	const int vw = fw - 2 * HWB, vh = fh - 2 * HWB;
	for (int pass = 0; pass < 2 && ok; ++pass) {
		void* dst = pass == 0 ? (void*)disparity : (void*)cost;
		if (!dst || (pass == 1 && !withCost)) continue;
		for (int y = 0; y < vh && ok; ++y) {
			fseek(f, dataStart + (long)pass * need + ((long)(y + HWB) * fw + HWB) * 2, SEEK_SET);
			ok = fread((char*)dst + (size_t)y * vw * 2, 2, (size_t)vw, f) == (size_t)vw;
		}
	}
	fclose(f);
	return ok ? 0 : -2;
}

} // extern "C"
