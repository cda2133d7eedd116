// ref_shim/ref_coslam_standin.cpp -- what src/app/SL_CoSLAM.cpp needs at link time from the GUI it is normally built into:
// the BA mutex and flags of MyApp (src/gui/MyApp.h:38-60) and the two redraw hooks (src/gui/CoSLAMThread.h:25-26).  Only the
// mutex is reachable from the code the tests drive (CoSLAM::~CoSLAM -> enterBACriticalSection).  TEST INFRASTRUCTURE.
#include "tools/GUI_ImageViewer.h"

pthread_mutex_t MyApp::s_mutexBA = PTHREAD_MUTEX_INITIALIZER;
bool MyApp::bBusyBAing = false;
bool MyApp::bCancelBA = false;
bool MyApp::bStop = false;
void updateDisplayData() {}
void redrawAllViews() {}
