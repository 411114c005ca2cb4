/* dmapio.h -- byte-exact reader/writer of the OpenMVS depth-data file ("depthNNNN.dmap").
 *
 * Replaces ExportDepthDataRaw / ImportDepthDataRaw (libs/MVS/DepthMap.cpp:1874-2037) and the
 * write-to-.tmp-then-rename of DepthData::Save (libs/MVS/DepthMap.cpp:234-252); header layout =
 * HeaderDepthDataRaw (libs/MVS/Interface.h:773-792), 28 bytes, little endian:
 *   u16 'DR' | u8 type (1 depth, 2 normal, 4 conf, 8 views) | u8 pad | u32 imageW,imageH,depthW,depthH | f32 dMin,dMax
 *   u16 nameLen, char name[] | u32 nIDs, u32 IDs[] | f64 K[9], R[9], C[3] | f32 depth[] | f32 normal[][3] | f32 conf[] | u8 views[][4]
 * The image file name is stored exactly as given (the reference stores it relative to the .dmap's folder).
 */
#ifndef DMAPIO_H_
#define DMAPIO_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct DMapHeader {
	uint32_t imageWidth, imageHeight, depthWidth, depthHeight;
	float dMin, dMax;
	uint32_t type;        /* bit mask of the planes present */
	uint32_t nIDs;        /* reference view ID followed by neighbour view IDs */
	uint32_t IDs[256];
	double K[9], R[9], C[3];
	char imageFileName[1024];
} DMapHeader;

/* normal / conf / views may be NULL (plane omitted).  Returns 0 on success. */
int dmap_write(const char* fileName, const DMapHeader* hdr, const float* depth, const float* normal, const float* conf, const uint8_t* views);
/* Fills hdr; returns 0 on success, -1 cannot open, -2 invalid file. */
int dmap_read_header(const char* fileName, DMapHeader* hdr);
/* Reads the planes selected by `flags` (ImportDepthDataRaw's flags, default 15) into caller buffers
 * sized from the header; a NULL pointer skips that plane. */
int dmap_read(const char* fileName, DMapHeader* hdr, unsigned flags, float* depth, float* normal, float* conf, uint8_t* views);

/* ---- .dimap: disparity-data file of the SGM path (SemiGlobalMatcher::ExportDisparityDataRawFull / ImportDisparityDataRawFull,
 * libs/MVS/SemiGlobalMatcher.cpp:2094-2188): i32 imageW, imageH | f64 H[9] | f64 Q[16] | i16 subpixelSteps | i32 cols, rows | i16 disparity[] | optional u16 cost[].
 * The maps are stored with the 3-pixel border (NO_DISP 32767 / NO_ACCUMCOST 65535) around the valid grid; these functions take / return the
 * valid grid (w x h) like the *Full variants.  cost may be NULL. */
int dimap_write(const char* fileName, int imageW, int imageH, const double H[9], const double Q[16], int16_t subpixelSteps,
                const int16_t* disparity, const uint16_t* cost, int w, int h);
/* Call with disparity == NULL to obtain the sizes and *hasCost, then with buffers of w*h entries (cost NULL skips it). */
int dimap_read(const char* fileName, int* imageW, int* imageH, double H[9], double Q[16], int16_t* subpixelSteps, int* w, int* h, int* hasCost,
               int16_t* disparity, uint16_t* cost);

#ifdef __cplusplus
}
#endif
#endif
