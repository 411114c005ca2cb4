/* optdense.h -- the reference's dense-reconstruction option table (namespace OPTDENSE, libs/MVS/DepthMap.cpp:67-114) as plain data, and its configuration file
 * (`DensifyPointCloud --dense-config-file`, apps/DensifyPointCloud/DensifyPointCloud.cpp:236-255).  Part of libmvsfront.so (host code, no GPU).
 *
 * The reference keeps every option twice: as a variable and as a string in the table OPTDENSE::oConfig, keyed by the option's TITLE ("Min Views Trust Point").
 * OPTDENSE::init() sets variable and string to the default; oConfig.Load(file) overwrites the strings from a text file of "Title = value" lines (an SML document,
 * libs/Common/SML.cpp:94-227; unknown titles are kept in the table and ignored); OPTDENSE::update() parses every string into its variable with `istream >> value`
 * (libs/Common/Common.h:124-131, libs/Common/Strings.h:160-163); the command-line flags are assigned afterwards.  When the file could not be read the reference writes
 * the table out to that path (DensifyPointCloud.cpp:253-254), which is how a template is obtained.
 * Here: mvsf_optdense_init = init(); mvsf_optdense_load = Load + update() (-2 and defaults if the file cannot be read, like bValidConfig == false);
 * mvsf_optdense_save = Save (one "Title = value" line per option, floats in the shortest form that reads back to the same value).
 * The struct's fields are the reference's variables, in the reference's order, under the reference's names; bool options are int32 (0 / 1).
 */
#ifndef OPTDENSE_H_
#define OPTDENSE_H_
#include <stdint.h>
#include "mvsfront.h"
#include "pmhip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct MVSFOptDense {
	uint32_t nResolutionLevel, nMaxResolution, nMinResolution, nSubResolutionLevels;
	uint32_t nMinViews, nMaxViews, nMinViewsFuse, nMinViewsFilter, nMinViewsFilterAdjust, nMinViewsTrustPoint, nNumViews, nPointInsideROI;
	int32_t bFilterAdjust, bAddCorners, bInitSparse, bRemoveDmaps;
	float fViewMinScore, fViewMinScoreRatio, fMinArea, fMinAngle, fOptimAngle, fMaxAngle;
	float fDescriptorMinMagnitudeThreshold, fDepthDiffThreshold, fNormalDiffThreshold, fPairwiseMul, fOptimizerEps;
	int32_t nOptimizerMaxIters;
	uint32_t nSpeckleSize, nIpolGapSize;
	int32_t nIgnoreMaskLabel;
	uint32_t nOptimize, nEstimateColors, nEstimateNormals;
	float fNCCThresholdKeep;
	uint32_t nEstimationIters, nEstimationGeometricIters;
	float fEstimationGeometricWeight;
	uint32_t nRandomIters, nRandomMaxScale;
	float fRandomDepthRatio, fRandomAngle1Range, fRandomAngle2Range, fRandomSmoothDepth, fRandomSmoothNormal, fRandomSmoothBonus;
} MVSFOptDense;

/* the table: number of options; option i's variable name, title, type ("uint32", "int32", "bool", "float") and default as the reference spells it */
int mvsf_optdense_count(void);
int mvsf_optdense_describe(int i, const char** name, const char** title, const char** type, const char** defval);
/* OPTDENSE::init(): every option at its default */
void mvsf_optdense_init(MVSFOptDense* o);
/* one option by title, from / to text.  set: 0, or -1 for a title the table does not have (the reference keeps such an entry and never reads it).
 * get: 0, -1 unknown title; value is NUL-terminated within cap. */
int mvsf_optdense_set(MVSFOptDense* o, const char* title, const char* value);
int mvsf_optdense_get(const MVSFOptDense* o, const char* title, char* value, int cap);
/* init() + Load(path) + update(): 0, or -2 when the file is not a valid configuration (bValidConfig == false: it cannot be opened -- o is at the defaults then --
 * or the document is malformed -- the entries in front of the error have been applied, as in the reference); *nUnknown (nullable) = entries whose title is not an option */
int mvsf_optdense_load(const char* path, MVSFOptDense* o, int* nUnknown);
int mvsf_optdense_save(const char* path, const MVSFOptDense* o);
/* the subsets the front end (mvsfront.h) and the estimator (pmhip.h) read; p->seed is left alone (the reference seeds from random_device) */
void mvsf_optdense_front(const MVSFOptDense* o, MVSFOptions* f);
void mvsf_optdense_estimator(const MVSFOptDense* o, PMHipParams* p);

#ifdef __cplusplus
}
#endif
#endif /* OPTDENSE_H_ */
