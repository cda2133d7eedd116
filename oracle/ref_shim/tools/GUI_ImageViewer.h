/* ref_shim/tools/GUI_ImageViewer.h -- stand-in (see math/SL_Matrix.h).  The GUI side of SL_CoSLAM.cpp: the reference's own
 * gui/MyApp.h is switched off through its include guard (-DMYAPP_H_, it needs wxWidgets); what SL_CoSLAM.cpp names of it --
 * the BA mutex and the three flags (src/gui/MyApp.h:38-60), updateDisplayData / redrawAllViews (src/gui/CoSLAMThread.h:25-26)
 * -- is declared here and defined in ref_coslam_standin.cpp. */
#ifndef REF_SHIM_GUI_IMAGEVIEWER_H
#define REF_SHIM_GUI_IMAGEVIEWER_H
#include <pthread.h>
class MyApp {
public:
    static pthread_mutex_t s_mutexBA;
    static bool bBusyBAing;
    static bool bCancelBA;
    static bool bStop;
};
void updateDisplayData();
void redrawAllViews();
#endif
