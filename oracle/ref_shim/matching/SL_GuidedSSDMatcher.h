/* ref_shim stand-in (see math/SL_Matrix.h): not on the call path */
