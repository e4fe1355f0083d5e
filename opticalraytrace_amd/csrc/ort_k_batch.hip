// ort_k_batch.hip — placeholder of the multi-system launches: filled in by the batch entry.
#include "ort_launch.h"
