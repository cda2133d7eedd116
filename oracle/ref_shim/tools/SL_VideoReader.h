/* ref_shim/tools/SL_VideoReader.h -- stand-in (see math/SL_Matrix.h): SingleSLAM holds a VideoReader*; nothing on the
 * tracker / BA call path decodes video. */
#ifndef REF_SHIM_SL_VIDEOREADER_H
#define REF_SHIM_SL_VIDEOREADER_H
class VideoReader {
public:
    int _w, _h;
    VideoReader() : _w(0), _h(0) {}
    virtual ~VideoReader() {}
    virtual void open(const char*) {}
    virtual void open() {}
    virtual void grabFrame() {}
    virtual void readCurFrame(unsigned char*, unsigned char*) {}
    virtual void getCurGrayImage(unsigned char*) {}
    virtual void getCurRGBImage(unsigned char*) {}
    virtual int getTotalFrame() { return 0; }
    virtual void skip(int) {}
};
#endif
