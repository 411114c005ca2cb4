// opt_dense.cpp -- see include/optdense.h.  The table below restates the option list of libs/MVS/DepthMap.cpp:69-114 (variable, title, first default): it has to
// match the reference entry for entry, and tests/test_optdense.py checks it against that file's text whenever /root/reference is present.
#include "../../include/optdense.h"
#include "sml_text.h"
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <sstream>
#include <string>

namespace {
enum Kind { U32, I32, BOOL, F32 };
struct Entry { const char* name; const char* title; Kind kind; const char* defval; size_t offset; };
#define OD(kind, name, title, def) {#name, title, kind, def, offsetof(MVSFOptDense, name)}
const Entry kTable[] = {
	OD(U32, nResolutionLevel, "Resolution Level", "1"),
	OD(U32, nMaxResolution, "Max Resolution", "3200"),
	OD(U32, nMinResolution, "Min Resolution", "640"),
	OD(U32, nSubResolutionLevels, "SubResolution levels", "2"),
	OD(U32, nMinViews, "Min Views", "2"),
	OD(U32, nMaxViews, "Max Views", "12"),
	OD(U32, nMinViewsFuse, "Min Views Fuse", "2"),
	OD(U32, nMinViewsFilter, "Min Views Filter", "2"),
	OD(U32, nMinViewsFilterAdjust, "Min Views Filter Adjust", "1"),
	OD(U32, nMinViewsTrustPoint, "Min Views Trust Point", "2"),
	OD(U32, nNumViews, "Num Views", "0"),
	OD(U32, nPointInsideROI, "Point Inside ROI", "1"),
	OD(BOOL, bFilterAdjust, "Filter Adjust", "1"),
	OD(BOOL, bAddCorners, "Add Corners", "0"),
	OD(BOOL, bInitSparse, "Init Sparse", "1"),
	OD(BOOL, bRemoveDmaps, "Remove Dmaps", "0"),
	OD(F32, fViewMinScore, "View Min Score", "2.0"),
	OD(F32, fViewMinScoreRatio, "View Min Score Ratio", "0.03"),
	OD(F32, fMinArea, "Min Area", "0.05"),
	OD(F32, fMinAngle, "Min Angle", "3.0"),
	OD(F32, fOptimAngle, "Optim Angle", "12.0"),
	OD(F32, fMaxAngle, "Max Angle", "65.0"),
	OD(F32, fDescriptorMinMagnitudeThreshold, "Descriptor Min Magnitude Threshold", "0.02"),
	OD(F32, fDepthDiffThreshold, "Depth Diff Threshold", "0.01"),
	OD(F32, fNormalDiffThreshold, "Normal Diff Threshold", "25"),
	OD(F32, fPairwiseMul, "Pairwise Mul", "0.3"),
	OD(F32, fOptimizerEps, "Optimizer Eps", "0.001"),
	OD(I32, nOptimizerMaxIters, "Optimizer Max Iters", "80"),
	OD(U32, nSpeckleSize, "Speckle Size", "100"),
	OD(U32, nIpolGapSize, "Interpolate Gap Size", "7"),
	OD(I32, nIgnoreMaskLabel, "Ignore Mask Label", "-1"),
	OD(U32, nOptimize, "Optimize", "7"),
	OD(U32, nEstimateColors, "Estimate Colors", "2"),
	OD(U32, nEstimateNormals, "Estimate Normals", "0"),
	OD(F32, fNCCThresholdKeep, "NCC Threshold Keep", "0.9"),
	OD(U32, nEstimationIters, "Estimation Iters", "3"),
	OD(U32, nEstimationGeometricIters, "Estimation Geometric Iters", "2"),
	OD(F32, fEstimationGeometricWeight, "Estimation Geometric Weight", "0.1"),
	OD(U32, nRandomIters, "Random Iters", "6"),
	OD(U32, nRandomMaxScale, "Random Max Scale", "2"),
	OD(F32, fRandomDepthRatio, "Random Depth Ratio", "0.003"),
	OD(F32, fRandomAngle1Range, "Random Angle1 Range", "16.0"),
	OD(F32, fRandomAngle2Range, "Random Angle2 Range", "10.0"),
	OD(F32, fRandomSmoothDepth, "Random Smooth Depth", "0.02"),
	OD(F32, fRandomSmoothNormal, "Random Smooth Normal", "13"),
	OD(F32, fRandomSmoothBonus, "Random Smooth Bonus", "0.93"),
};
#undef OD
const int kCount = (int)(sizeof(kTable) / sizeof(kTable[0]));
const char* const kKindName[] = {"uint32", "int32", "bool", "float"};

const Entry* findTitle(const char* title) {
	for (const Entry& e : kTable) if (strcmp(e.title, title) == 0) return &e;
	return nullptr;
}
// String::FromString (libs/Common/Strings.h:160-163): `istringstream >> value`, whatever the text
template <typename T> T fromString(const char* text, T start) { T v(start); std::istringstream is(text); is >> v; return v; }
void assign(MVSFOptDense* o, const Entry& e, const char* text) {
	char* at = (char*)o + e.offset;
	switch (e.kind) {
	case U32: *(uint32_t*)at = fromString<uint32_t>(text, *(uint32_t*)at); break;
	case I32: *(int32_t*)at = fromString<int32_t>(text, *(int32_t*)at); break;
	case BOOL: *(int32_t*)at = fromString<bool>(text, *(int32_t*)at != 0) ? 1 : 0; break;
	case F32: *(float*)at = fromString<float>(text, *(float*)at); break;
	}
}
std::string toText(const MVSFOptDense* o, const Entry& e) {
	const char* at = (const char*)o + e.offset;
	char buf[64];
	switch (e.kind) {
	case U32: snprintf(buf, sizeof(buf), "%u", (unsigned)*(const uint32_t*)at); break;
	case I32: snprintf(buf, sizeof(buf), "%d", (int)*(const int32_t*)at); break;
	case BOOL: snprintf(buf, sizeof(buf), "%d", *(const int32_t*)at != 0 ? 1 : 0); break;
	case F32: {
		const float v = *(const float*)at;
		for (int digits = 1; digits <= 9; ++digits) {      // the shortest decimal that reads back as v
			snprintf(buf, sizeof(buf), "%.*g", digits, (double)v);
			if (fromString<float>(buf, 0.f) == v) break;
		}
		break; }
	}
	return std::string(buf);
}
} // namespace

extern "C" {

int mvsf_optdense_count(void) { return kCount; }
int mvsf_optdense_describe(int i, const char** name, const char** title, const char** type, const char** defval) {
	if (i < 0 || i >= kCount) return -1;
	if (name) *name = kTable[i].name;
	if (title) *title = kTable[i].title;
	if (type) *type = kKindName[kTable[i].kind];
	if (defval) *defval = kTable[i].defval;
	return 0;
}
void mvsf_optdense_init(MVSFOptDense* o) {
	if (!o) return;
	memset(o, 0, sizeof(*o));
	for (const Entry& e : kTable) assign(o, e, e.defval);
}
int mvsf_optdense_set(MVSFOptDense* o, const char* title, const char* value) {
	if (!o || !title || !value) return -1;
	const Entry* e = findTitle(title);
	if (!e) return -1;
	assign(o, *e, value);
	return 0;
}
int mvsf_optdense_get(const MVSFOptDense* o, const char* title, char* value, int cap) {
	if (!o || !title || !value || cap <= 0) return -1;
	const Entry* e = findTitle(title);
	if (!e) return -1;
	const std::string t = toText(o, *e);
	strncpy(value, t.c_str(), (size_t)cap - 1); value[cap - 1] = 0;
	return 0;
}
int mvsf_optdense_load(const char* path, MVSFOptDense* o, int* nUnknown) {
	if (!o) return -1;
	mvsf_optdense_init(o);
	if (nUnknown) *nUnknown = 0;
	if (!path) return -1;
	try {
		std::vector<std::pair<std::string, std::string>> items;
		const int rd = smltext::rootValues(path, items);
		if (rd < 0) return -2;
		int unknown = 0;
		for (const auto& it : items) {      // a title given twice: the later line wins (one table entry per title)
			const Entry* e = it.first.empty() ? nullptr : findTitle(it.first.c_str());
			if (e) assign(o, *e, it.second.c_str()); else ++unknown;
		}
		if (nUnknown) *nUnknown = unknown;
		return rd == 0 ? 0 : -2;            // a malformed document is "not valid" (bValidConfig), but what was read in front of the error has been applied, as there
	} catch (...) { mvsf_optdense_init(o); return -2; }
}
int mvsf_optdense_save(const char* path, const MVSFOptDense* o) {
	if (!path || !o) return -1;
	FILE* f = fopen(path, "wb");
	if (!f) return -2;
	for (const Entry& e : kTable) fprintf(f, "%s = %s\n", e.title, toText(o, e).c_str());   // SML::SaveIntern's line format (SML.cpp:266-268); the reference's line order is a hash map's
	return fclose(f) == 0 ? 0 : -2;
}
void mvsf_optdense_front(const MVSFOptDense* o, MVSFOptions* f) {
	if (!o || !f) return;
	f->nMinViews = o->nMinViews; f->nMaxViews = o->nMaxViews; f->nMinViewsTrustPoint = o->nMinViewsTrustPoint; f->nNumViews = o->nNumViews; f->nPointInsideROI = o->nPointInsideROI;
	f->fViewMinScore = o->fViewMinScore; f->fViewMinScoreRatio = o->fViewMinScoreRatio; f->fMinArea = o->fMinArea; f->fMinAngle = o->fMinAngle;
	f->fOptimAngle = o->fOptimAngle; f->fMaxAngle = o->fMaxAngle;
}
void mvsf_optdense_estimator(const MVSFOptDense* o, PMHipParams* p) {
	if (!o || !p) return;
	p->nSubResolutionLevels = o->nSubResolutionLevels; p->nEstimationIters = o->nEstimationIters; p->nEstimationGeometricIters = o->nEstimationGeometricIters;
	p->nRandomIters = o->nRandomIters; p->fEstimationGeometricWeight = o->fEstimationGeometricWeight; p->fRandomDepthRatio = o->fRandomDepthRatio;
	p->fRandomAngle1Range = o->fRandomAngle1Range; p->fRandomAngle2Range = o->fRandomAngle2Range; p->fRandomSmoothDepth = o->fRandomSmoothDepth;
	p->fRandomSmoothNormal = o->fRandomSmoothNormal; p->fRandomSmoothBonus = o->fRandomSmoothBonus; p->fNCCThresholdKeep = o->fNCCThresholdKeep;
	p->fDescriptorMinMagnitudeThreshold = o->fDescriptorMinMagnitudeThreshold;
}

} // extern "C"
